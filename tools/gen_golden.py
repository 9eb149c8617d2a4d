#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python on CPU.

Runs only in the build container (needs /root/reference); the fixtures it writes
are plain data (inputs + expected outputs) and are committed, the reference is not.

How the reference is made importable:
  * cv2 / mcubes / trimesh / torch_efficient_distloss are imported at the top of
    nerf/renderer.py and nerf/utils.py but unused on this path -> empty stub modules;
  * encoding.get_encoder lazily imports `gridencoder`, `shencoder`, `freqencoder`
    (CUDA-only in the reference) -> modules backed by the CPU oracle are registered
    under those names, so the reference's renderer/network run unmodified on top of
    the oracle's encoder arithmetic.
Everything else (get_rays, near/far, contraction, sample_pdf + searchsorted,
MLPs, trunc_exp, compositing, SAM / mask heads, mask NLL) is the reference's code.

Model parameters are not stored: every tensor is regenerated from
(name, shape, seed, lo, hi) by sanerf_hq_amd.synth.hash_uniform, recorded in the
fixture's `param_spec` JSON.
"""
import argparse
import json
import os
import sys
import types
import zlib

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import oracle as orc  # noqa: E402
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("synth", os.path.join(ROOT, "sanerf-hq_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

GOLD = os.path.join(ROOT, "tests", "golden")


# --------------------------------------------------------------------------
# reference import harness
# --------------------------------------------------------------------------
class _GridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, embeddings, mod):
        x = inputs.detach().cpu().numpy()
        out, _ = orc.grid_encode_forward(x, embeddings.detach().cpu().numpy(), mod.offsets.numpy(),
                                         mod.per_level_scale, mod.base_resolution, False,
                                         mod.gridtype_id, mod.align_corners, mod.interp_id)
        ctx.mod = mod
        ctx.save_for_backward(inputs, embeddings)
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, grad):
        inputs, embeddings = ctx.saved_tensors
        mod = ctx.mod
        ge, _ = orc.grid_encode_backward(grad.contiguous().numpy(), inputs.detach().numpy(),
                                         embeddings.detach().numpy(), mod.offsets.numpy(),
                                         mod.per_level_scale, mod.base_resolution, None,
                                         mod.gridtype_id, mod.align_corners, mod.interp_id)
        return None, torch.from_numpy(ge), None


class OracleGridEncoder(nn.Module):
    """Constructor/forward contract of gridencoder/grid.py:102-168, arithmetic by the oracle."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype='hash', align_corners=False,
                 interpolation='linear'):
        super().__init__()
        offsets, pls = orc.grid_layout(input_dim, num_levels, level_dim, per_level_scale, base_resolution,
                                       log2_hashmap_size, desired_resolution)
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution = pls, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype_id = {'hash': 0, 'tiled': 1}[gridtype]
        self.interp_id = {'linear': 0, 'smoothstep': 1}[interpolation]
        self.align_corners = align_corners
        self.register_buffer('offsets', torch.from_numpy(offsets))
        self.embeddings = nn.Parameter(torch.empty(int(offsets[-1]), level_dim).uniform_(-1e-4, 1e-4))

    def forward(self, inputs, bound=1, max_level=None):
        inputs = (inputs + bound) / (2 * bound)
        prefix = list(inputs.shape[:-1])
        inputs = inputs.view(-1, self.input_dim)
        out = _GridFn.apply(inputs, self.embeddings, self)
        return out.view(prefix + [self.output_dim])


class OracleSHEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree, self.output_dim = input_dim, degree, degree ** 2

    def forward(self, inputs, size=1):
        inputs = inputs / size
        inputs = inputs / torch.norm(inputs, dim=-1, keepdim=True)
        prefix = list(inputs.shape[:-1])
        x = inputs.reshape(-1, self.input_dim).detach().numpy()
        out, _ = orc.sh_encode_forward(x, self.degree)
        return torch.from_numpy(out).reshape(prefix + [self.output_dim])


class OracleFreqEncoder(nn.Module):
    def __init__(self, input_dim=3, degree=4):
        super().__init__()
        self.input_dim, self.degree = input_dim, degree
        self.output_dim = input_dim + input_dim * 2 * degree

    def forward(self, inputs, **kw):
        prefix = list(inputs.shape[:-1])
        out = orc.freq_encode_forward(inputs.reshape(-1, self.input_dim).detach().numpy(), self.degree)
        return torch.from_numpy(out).reshape(prefix + [self.output_dim])


def install_reference():
    for name in ("cv2", "mcubes", "trimesh"):
        sys.modules.setdefault(name, types.ModuleType(name))
    ed = types.ModuleType("torch_efficient_distloss")
    ed.eff_distloss = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("eff_distloss not available"))
    sys.modules["torch_efficient_distloss"] = ed
    for modname, cls, attr in (("gridencoder", OracleGridEncoder, "GridEncoder"),
                               ("shencoder", OracleSHEncoder, "SHEncoder"),
                               ("freqencoder", OracleFreqEncoder, "FreqEncoder")):
        m = types.ModuleType(modname)
        setattr(m, attr, cls)
        sys.modules[modname] = m
    sys.path.insert(0, REF)
    import nerf.renderer as rr
    import nerf.network as nn_
    import nerf.utils as ru
    import encoding as enc
    return rr, nn_, ru, enc


def make_opt(**kw):
    opt = types.SimpleNamespace(
        bound=128, contract=True, min_near=0.2, density_thresh=10, render_mesh=False,
        num_steps=[128, 64, 32], with_mask=False, with_sam=False, n_inst=2, mask_mlp_type='default',
        background='last_sample', lambda_proposal=0.0, lambda_distort=0.0, max_ray_batch=16384,
        sam_use_view_direction=True, epsilon=1e-6, num_rays=4096)
    for k, v in kw.items():
        setattr(opt, k, v)
    return opt


# --------------------------------------------------------------------------
# deterministic parameters
# --------------------------------------------------------------------------
def param_spec_for(model, base_seed, table_amp, mlp_gain, level_decay=0.7):
    spec = []
    sd = model.state_dict()
    for name, p in sd.items():
        if name.endswith("offsets") or name.startswith("aabb"):
            continue
        seed = (zlib.crc32(name.encode()) ^ base_seed) & 0x7FFFFFFF
        shape = list(p.shape)
        extra = {}
        if name.endswith("embeddings"):
            lo, hi = -table_amp, table_amp
            offs = [int(v) for v in sd[name[:-len("embeddings")] + "offsets"].tolist()]
            extra = dict(offsets=offs, level_scale=[float(level_decay ** l) for l in range(len(offs) - 1)])
        elif name.endswith("weight") and p.dim() == 2:
            b = mlp_gain / np.sqrt(shape[1])
            lo, hi = -b, b
        elif name.endswith("weight") and p.dim() == 1:   # LayerNorm weight
            lo, hi = 0.5, 1.5
        else:                                            # biases
            lo, hi = -0.1, 0.1
        spec.append(dict(name=name, shape=shape, seed=int(seed), lo=float(lo), hi=float(hi), **extra))
    return spec


def load_params(model, spec):
    sd = model.state_dict()
    for s in spec:
        sd[s["name"]].copy_(torch.from_numpy(synth.make_param(s)))


# --------------------------------------------------------------------------
# capture helpers
# --------------------------------------------------------------------------
class Capture:
    """Records what the reference computes inside run() without re-typing it:
    sample_pdf inputs/outputs, the integer searchsorted result and the linspace tables."""

    def __init__(self, rr):
        self.rr = rr
        self.pdf_calls = []
        self.searchsorted = []
        self.linspace = []

    def __enter__(self):
        rr = self.rr
        self._sp = rr.sample_pdf
        self._ss = torch.searchsorted
        self._ls = torch.linspace

        def ss(*a, **k):
            r = self._ss(*a, **k)
            self.searchsorted.append(r.clone())
            return r

        def ls(*a, **k):
            r = self._ls(*a, **k)
            self.linspace.append(r.clone().cpu())
            return r

        def sp(bins, weights, T, perturb=False):
            n0 = len(self.searchsorted)
            l0 = len(self.linspace)
            out = self._sp(bins, weights, T, perturb)
            self.pdf_calls.append(dict(bins=bins.clone(), weights=weights.clone(), T=T, out=out.clone(),
                                       inds=self.searchsorted[n0].clone(), u=self.linspace[l0].clone()))
            return out

        rr.sample_pdf = sp
        torch.searchsorted = ss
        torch.linspace = ls
        return self

    def __exit__(self, *exc):
        self.rr.sample_pdf = self._sp
        torch.searchsorted = self._ss
        torch.linspace = self._ls


def subset_rays(ru, H, W, n_side, pose, device="cpu"):
    """Full-image rays from the reference's get_rays, then an n_side x n_side lattice of pixels."""
    fx, fy, cx, cy = synth.pinhole_intrinsics(H, W)
    intr = np.array([fx, fy, cx, cy], dtype=np.float32)
    res = ru.get_rays(torch.from_numpy(pose)[None], intr, H, W, -1)
    ro, rd = res["rays_o"][0] if res["rays_o"].dim() == 3 else res["rays_o"], res["rays_d"]
    ro = ro.reshape(-1, 3); rd = rd.reshape(-1, 3)
    ys = np.linspace(0, H - 1, n_side).round().astype(np.int64)
    xs = np.linspace(0, W - 1, n_side).round().astype(np.int64)
    idx = (ys[:, None] * W + xs[None, :]).reshape(-1)
    return ro[idx].contiguous().clone(), rd[idx].contiguous().clone(), idx


def np_(t):
    return t.detach().cpu().numpy()


def compare(name, a, b, tol=None, exact=False):
    a = np.asarray(a); b = np.asarray(b)
    if exact:
        bad = int((a != b).sum())
        print(f"   {name:28s} mismatches {bad}/{a.size}")
        return bad
    fin = np.isfinite(a) & np.isfinite(b)
    d = np.abs(a[fin].astype(np.float64) - b[fin].astype(np.float64))
    m = float(d.max()) if d.size else 0.0
    print(f"   {name:28s} max|diff| {m:.3e}  (finite {fin.sum()}/{a.size})" + ("" if tol is None or m <= tol else "   <-- ABOVE TOL"))
    return m


# --------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------
def fx_units(rr, ru, enc):
    print("[units] get_rays / near_far / contract / sample_pdf / freq / linspace")
    out = {}
    # get_rays (nerf/utils.py:182-304): small image + known-answer pose
    for tag, (H, W, pose) in dict(a=(4, 6, np.eye(4, dtype=np.float32)),
                                  b=(16, 24, synth.orbit_pose(1.3, 35.0, -50.0))).items():
        fx, fy, cx, cy = synth.pinhole_intrinsics(H, W)
        res = ru.get_rays(torch.from_numpy(pose)[None], np.array([fx, fy, cx, cy], dtype=np.float32), H, W, -1)
        ro, rd = np_(res["rays_o"]).reshape(-1, 3), np_(res["rays_d"]).reshape(-1, 3)
        out[f"rays_{tag}_pose"] = pose
        out[f"rays_{tag}_intr"] = np.array([fx, fy, cx, cy, H, W], dtype=np.float64)
        out[f"rays_{tag}_o"], out[f"rays_{tag}_d"] = ro, rd
        o2, d2 = orc.generate_rays(pose, fx, fy, cx, cy, H, W)
        compare(f"get_rays[{tag}] rays_d", rd, d2, 1e-6); compare(f"get_rays[{tag}] rays_o", ro, o2, 0)
        if tag == "b":   # pixel-subset branch (utils.py:211-212, 262-268, 294-300): coords = (row, col) pairs
            coords = torch.tensor([[0, 0], [3, 5], [15, 23], [7, 11], [2, 20], [15, 0]])
            sub = ru.get_rays(torch.from_numpy(pose)[None], np.array([fx, fy, cx, cy], dtype=np.float32), H, W, coords.shape[0], coords=coords)
            out["rays_b_coords"] = np_(coords).astype(np.int64)
            out["rays_b_sub_o"], out["rays_b_sub_d"] = np_(sub["rays_o"]).reshape(-1, 3), np_(sub["rays_d"]).reshape(-1, 3)
            out["rays_b_sub_i"], out["rays_b_sub_j"] = np_(sub["i"]).reshape(-1).astype(np.int64), np_(sub["j"]).reshape(-1).astype(np.int64)
            out["rays_b_sub_inds_coarse"] = np_(sub["inds_coarse"]).reshape(-1).astype(np.int64)
    # near_far (renderer.py:122-139), incl. rays that miss the box and axis-parallel rays
    g = torch.Generator().manual_seed(1)
    ro = (torch.rand(512, 3, generator=g) - 0.5) * 3
    rd = torch.randn(512, 3, generator=g)
    rd[:16, 0] = 0.0
    ro[16:32] = ro[16:32] * 200          # outside a +-128 box
    for tag, aabb in (("big", [-128.0] * 3 + [128.0] * 3), ("small", [-1.0, -0.5, -1.0, 1.0, 0.7, 1.2])):
        ab = torch.tensor(aabb)
        n, f = rr.near_far_from_aabb(ro, rd, ab, 0.2)
        out[f"nf_{tag}_aabb"], out[f"nf_{tag}_near"], out[f"nf_{tag}_far"] = np_(ab), np_(n), np_(f)
        n2, f2 = orc.near_far_from_aabb(np_(ro), np_(rd), np_(ab), 0.2)
        compare(f"near[{tag}]", np_(n), n2, 0); compare(f"far[{tag}]", np_(f), f2, 0)
    out["nf_o"], out["nf_d"] = np_(ro), np_(rd)
    # contract (renderer.py:60-69)
    x = (torch.rand(4096, 3, generator=g) - 0.5) * torch.tensor([1.0, 4.0, 40.0])
    x[:4] = torch.tensor([[0.5, -0.5, 0.25], [1.0, 1.0, 0.5], [-3.0, 3.0, 1.0], [0.0, 0.0, 0.0]])
    z = rr.contract(x.clone())
    out["contract_x"], out["contract_z"] = np_(x), np_(z)
    compare("contract", np_(z), orc.contract(np_(x)), 0)
    # sample_pdf (renderer.py:84-119) on synthetic weights: spiky, flat, zero, and one-hot rows
    for tag, (T0, T) in dict(a=(128, 65), b=(64, 33), c=(32, 17), d=(48, 33)).items():
        N = 64
        w = torch.rand(N, T0, generator=g) ** 8
        w[0] = 0.0; w[1] = 1.0 / T0; w[2] = 0.0; w[2, T0 // 3] = 1.0; w[3] = 0.0; w[3, -1] = 1.0
        b = torch.sort(torch.rand(N, T0 + 1, generator=g), dim=-1).values
        b[4] = torch.linspace(0, 1, T0 + 1)
        with Capture(rr) as cap:
            o = rr.sample_pdf(b, w, T, False)
        c = cap.pdf_calls[0]
        out[f"pdf_{tag}_bins"], out[f"pdf_{tag}_w"], out[f"pdf_{tag}_out"] = np_(b), np_(w), np_(o)
        out[f"pdf_{tag}_inds"], out[f"pdf_{tag}_u"] = np_(c["inds"]).astype(np.int32), np_(c["u"])
        o2, i2 = orc.sample_pdf(np_(b), np_(w), T, u=np_(c["u"]))
        compare(f"sample_pdf[{tag}] inds", np_(c["inds"]), i2, exact=True)
        compare(f"sample_pdf[{tag}] bins", np_(o), o2, 1e-6)
        o3, i3 = orc.sample_pdf(np_(b), np_(w), T)
        compare(f"sample_pdf[{tag}] inds(own u)", np_(c["inds"]), i3, exact=True)
        compare(f"  u table vs orc_linspace", np_(c["u"]), orc.linspace(0.5 / T, 1 - 0.5 / T, T), exact=True)
    # frequency encoding vs FreqEncoder_torch (encoding.py:6-44)
    for deg in (4, 6, 10):
        fe, _ = enc.get_encoder("frequency_torch", input_dim=3, multires=deg)
        xin = (torch.rand(257, 3, generator=g) - 0.5) * 2
        y = fe(xin)
        out[f"freq{deg}_x"], out[f"freq{deg}_y"] = np_(xin), np_(y)
        compare(f"freq deg={deg}", np_(y), orc.freq_encode_forward(np_(xin), deg), 2e-6)
    # linspace tables used by run()/sample_pdf
    for steps in (129, 65, 33, 17, 49, 97):
        out[f"linspace01_{steps}"] = np_(torch.linspace(0, 1, steps))
        compare(f"linspace(0,1,{steps})", out[f"linspace01_{steps}"], orc.linspace(0, 1, steps), exact=True)
    np.savez_compressed(os.path.join(GOLD, "units.npz"), **out)


def build_cfg_from_model(model, opt, keep):
    """orc_render_cfg from a reference NeRFNetwork instance (weights read from its state_dict)."""
    cfg = orc.OrcRenderCfg()
    S = len(opt.num_steps)
    cfg.num_stages = S
    for k, t in enumerate(opt.num_steps):
        cfg.num_steps[k] = t

    def grid_of(enc_):
        return orc.make_grid(np_(enc_.embeddings), np_(enc_.offsets), enc_.per_level_scale, enc_.base_resolution,
                             enc_.input_dim, enc_.gridtype_id, enc_.align_corners, enc_.interp_id, keep=keep)

    def mlp_of(m, act="relu", skip=()):
        ws = [np_(l.weight) for l in m.net]
        bs = [np_(l.bias) if l.bias is not None else None for l in m.net]
        return orc.make_mlp(ws, bs, act, skip, keep=keep, dim_in=m.dim_in)

    for k in range(S - 1):
        cfg.prop_grid[k] = grid_of(model.prop_encoders[k])
        cfg.prop_mlp[k] = mlp_of(model.prop_mlp[k])
    cfg.grid = grid_of(model.grid)
    cfg.grid_mlp = mlp_of(model.grid_mlp)
    cfg.view_mlp = mlp_of(model.view_mlp)
    cfg.sh_degree = model.view_encoder.degree
    ab = np_(model.aabb_infer)
    for i in range(6):
        cfg.aabb[i] = float(ab[i])
    cfg.min_near = model.min_near
    cfg.bound = float(model.bound)
    cfg.contract = int(opt.contract)
    cfg.last_sample_opaque = int(opt.background == 'last_sample')
    cfg.bg_color = 1.0
    if opt.with_sam:
        cfg.with_sam = 1
        cfg.s_grid = grid_of(model.s_grid)
        cfg.samvit_mlp = mlp_of(model.samvit_mlp[0], "leaky", model.samvit_mlp[0].skip_layers)
        lw = keep.hold(np_(model.samvit_mlp[1].weight).copy()); lb = keep.hold(np_(model.samvit_mlp[1].bias).copy())
        cfg.ln_weight, cfg.ln_bias, cfg.ln_eps = lw.ctypes.data, lb.ctypes.data, model.samvit_mlp[1].eps
    if opt.with_mask:
        cfg.with_mask = 1
        cfg.m_grid = grid_of(model.m_grid)
        cfg.mask_mlp = mlp_of(model.mask_mlp[0], "leaky", model.mask_mlp[0].skip_layers)
    return cfg


def fx_render(rr, nn_, ru, tag, num_steps, with_heads, n_side, H, W, base_seed, table_amp, mlp_gain,
              pose=None, time_it=False, tables_f16=False):
    print(f"[render:{tag}] num_steps={num_steps} heads={with_heads} rays={n_side * n_side}")
    opt = make_opt(num_steps=num_steps, with_sam=with_heads, with_mask=with_heads)
    torch.manual_seed(0)
    model = nn_.NeRFNetwork(opt)
    spec = param_spec_for(model, base_seed, table_amp, mlp_gain)
    load_params(model, spec)
    if tables_f16:     # the reference computes in fp32 on tables whose VALUES are fp16-representable: what a half-precision table copy holds
        with torch.no_grad():
            for name, p_ in model.named_parameters():
                if name.endswith("embeddings"):
                    p_.copy_(p_.half().float())
    pose = synth.orbit_pose(1.0, 20.0, 30.0) if pose is None else pose
    ro, rd, idx = subset_rays(ru, H, W, n_side, pose)
    N = ro.shape[0]

    sig_last = {}

    def hook(mod, args, outp):
        if isinstance(outp, dict) and "color" in outp:
            sig_last["sigma"] = outp["sigma"].detach().clone()
            sig_last["xyzs"] = args[0].detach().clone()
    h = model.register_forward_hook(hook)
    dens = []
    _density = model.density

    def density(x, proposal=-1):
        r = _density(x, proposal=proposal)
        dens.append(r["sigma"].detach().clone())
        return r
    model.density = density

    # train() makes run() return results['weights'] for the RGB-only model (renderer.py:342-345)
    model.train(not with_heads)
    with torch.no_grad(), Capture(rr) as cap:
        kw = dict(return_feats=1, return_mask=1, H=n_side, W=n_side) if with_heads else {}
        res = model.render(ro, rd, staged=False, perturb=False, update_proposal=False, **kw)
    h.remove()
    model.eval()

    keep = orc._Keep()
    cfg = build_cfg_from_model(model, opt, keep)
    u_tables = {k + 1: np_(c["u"]) for k, c in enumerate(cap.pdf_calls)}
    got = orc.render(cfg, np_(ro), np_(rd), debug=True, u_tables=u_tables,
                     bins0_table=np_(torch.linspace(0, 1, num_steps[0] + 1)))

    out = dict(rays_o=np_(ro), rays_d=np_(rd), pixel_index=idx.astype(np.int64),
               image=np_(res["image"]), depth=np_(res["depth"]), weights_sum=np_(res["weights_sum"]),
               param_spec=np.array(json.dumps(spec)), num_steps=np.array(num_steps, dtype=np.int64),
               with_heads=np.array(int(with_heads)), HW=np.array([H, W, n_side], dtype=np.int64), pose=pose,
               tables_f16=np.array(int(tables_f16)))
    for k, c in enumerate(cap.pdf_calls):
        out[f"bins{k}"] = np_(c["bins"]); out[f"weights{k}"] = np_(c["weights"])
        out[f"inds{k + 1}"] = np_(c["inds"]).astype(np.int32); out[f"u{k + 1}"] = np_(c["u"])
        out[f"sigmas{k}"] = np_(dens[k])
    S = len(num_steps)
    out[f"bins{S - 1}"] = np_(cap.pdf_calls[-1]["out"]) if S > 1 else np_(torch.linspace(0, 1, num_steps[0] + 1).expand(N, -1))
    out[f"sigmas{S - 1}"] = np_(sig_last["sigma"])
    out["xyzs_last"] = np_(sig_last["xyzs"])
    if "weights" in res:
        out[f"weights{S - 1}"] = np_(res["weights"])
    if with_heads:
        out["samvit"] = np_(res["samvit"]).reshape(N, -1)
        out["instance_mask_logits"] = np_(res["instance_mask_logits"])

    # --- oracle vs reference report (the oracle's pin) ---
    sg = out[f"sigmas{S - 1}"]
    print(f"   sigma(last) range [{sg.min():.3e}, {sg.max():.3e}]  median {np.median(sg):.3e};"
          f" weights_sum mean {out['weights_sum'].mean():.4f}")
    stats = {}
    for k in range(S):
        compare(f"bins{k}", out[f"bins{k}"], got[f"bins{k}"], 1e-6)
        compare(f"sigmas{k}", out[f"sigmas{k}"], got[f"sigmas{k}"])
        if f"weights{k}" in out:
            compare(f"weights{k}", out[f"weights{k}"], got[f"weights{k}"], 1e-5)
        if k >= 1:
            bad = compare(f"inds{k}", out[f"inds{k}"], got[f"inds{k}"], exact=True)
            stats[f"inds{k}_mismatch"] = bad
            stats[f"inds{k}_total"] = int(out[f"inds{k}"].size)
    compare("xyzs_last", out["xyzs_last"], got["xyzs_last"], 1e-5)
    for key, tol in (("image", 1e-4), ("depth", 1e-4), ("weights_sum", 1e-5)):
        stats[key] = compare(key, out[key], got[key], tol)
    if with_heads:
        stats["samvit"] = compare("samvit", out["samvit"], got["samvit"], 1e-4)
        stats["mask"] = compare("instance_mask_logits", out["instance_mask_logits"], got["instance_mask_logits"], 1e-4)
    out["oracle_vs_reference"] = np.array(json.dumps(stats))
    np.savez_compressed(os.path.join(GOLD, f"render_{tag}.npz"), **out)

    if time_it:
        import time
        t0 = time.time()
        with torch.no_grad():
            model.render(ro, rd, staged=False, perturb=False, update_proposal=False, **kw)
        t_ref = time.time() - t0
        t0 = time.time()
        orc.render(cfg, np_(ro), np_(rd))
        t_orc = time.time() - t0
        print(f"   timing on {N} rays: reference python {N / t_ref:.1f} rays/s, oracle {N / t_orc:.1f} rays/s"
              f" ({orc.num_threads()} threads)")
    return model, opt, spec


def fx_c1(rr, nn_, enc):
    """BASELINE config C1: 64x64, hashgrid L=8 T=2^14, 1-hidden-x32 sigma MLP, 32 samples/ray.
    The reference's NeRFNetwork hard-codes its sizes, so C1 is a small subclass of the reference's
    NeRFRenderer (unmodified run()) — SURVEY.md §8a."""
    print("[render:c1] 64x64, L=8 T=2^14, 1x32 MLP, 32 spp")
    opt = make_opt(num_steps=[32])

    class C1Field(rr.NeRFRenderer):
        def __init__(self, opt):
            super().__init__(opt)
            self.grid, d = enc.get_encoder("hashgrid", input_dim=3, level_dim=2, num_levels=8,
                                           log2_hashmap_size=14, desired_resolution=2048)
            self.grid_mlp = nn_.MLP(d, 16, 32, 2, bias=False)
            self.view_encoder, vd = enc.get_encoder("sh", input_dim=3, degree=4)
            self.view_mlp = nn_.MLP(15 + vd, 3, 32, 2, bias=False)

        def forward(self, x, d, **kw):
            f = self.grid_mlp(self.grid(x, bound=self.bound))
            sigma = nn_.trunc_exp(f[..., 0])
            return dict(sigma=sigma, geo_feat=f[..., 1:], color=torch.cat([f[..., 1:], self.view_encoder(d)], -1),
                        grid_output=None)

    model = C1Field(opt).eval()
    spec = param_spec_for(model, 77, 1.0, 4.0)
    load_params(model, spec)
    H = W = 64
    pose = synth.orbit_pose(1.0, 20.0, 30.0)
    fx, fy, cx, cy = synth.pinhole_intrinsics(H, W)
    import nerf.utils as ru
    res = ru.get_rays(torch.from_numpy(pose)[None], np.array([fx, fy, cx, cy], dtype=np.float32), H, W, -1)
    ro, rd = res["rays_o"].reshape(-1, 3).contiguous(), res["rays_d"].reshape(-1, 3).contiguous()
    import time
    t0 = time.time()
    with torch.no_grad():
        out_ref = model.render(ro, rd, staged=True, perturb=False)
    t_ref = time.time() - t0

    keep = orc._Keep()
    cfg = orc.OrcRenderCfg()
    cfg.num_stages = 1; cfg.num_steps[0] = 32
    g = model.grid
    cfg.grid = orc.make_grid(np_(g.embeddings), np_(g.offsets), g.per_level_scale, g.base_resolution, keep=keep)
    cfg.grid_mlp = orc.make_mlp([np_(l.weight) for l in model.grid_mlp.net], keep=keep)
    cfg.view_mlp = orc.make_mlp([np_(l.weight) for l in model.view_mlp.net], keep=keep)
    cfg.sh_degree = 4
    for i, v in enumerate(np_(model.aabb_infer)):
        cfg.aabb[i] = float(v)
    cfg.min_near, cfg.bound, cfg.contract, cfg.last_sample_opaque, cfg.bg_color = 0.2, 2.0, 1, 1, 1.0
    t0 = time.time()
    got = orc.render(cfg, np_(ro), np_(rd))
    t_orc = time.time() - t0
    stats = {k: compare(k, np_(out_ref[k]), got[k], 1e-4) for k in ("image", "depth", "weights_sum")}
    print(f"   C1 timing: reference python {H * W / t_ref:.0f} rays/s, oracle {H * W / t_orc:.0f} rays/s")
    np.savez_compressed(os.path.join(GOLD, "render_c1.npz"), image=np_(out_ref["image"]), depth=np_(out_ref["depth"]),
                        weights_sum=np_(out_ref["weights_sum"]), pose=pose, HW=np.array([H, W], dtype=np.int64),
                        param_spec=np.array(json.dumps(spec)), oracle_vs_reference=np.array(json.dumps(stats)),
                        ref_rays_per_s=np.array(H * W / t_ref))


def fx_train(rr, nn_, ru):
    """BASELINE config C5: mask-field training step (trainer.py:401-428,473): fwd+bwd of m_grid + mask_mlp,
    NLL of softmax(logits) clamped to [eps,1-eps], everything else frozen.  512 rays here (4096 in bench)."""
    print("[train:c5] mask-field NLL step, 512 rays")
    opt = make_opt(num_steps=[128, 64, 32], with_sam=False, with_mask=True)
    torch.manual_seed(0)
    model = nn_.NeRFNetwork(opt)
    spec = param_spec_for(model, 4242, 1.0, 4.0)
    load_params(model, spec)
    for n_, p in model.named_parameters():
        p.requires_grad_(n_.startswith("m_grid") or n_.startswith("mask_mlp"))   # main.py:249-256 freeze
    model.train()
    H = W = 512
    pose = synth.orbit_pose(1.1, 25.0, 60.0)
    fx, fy, cx, cy = synth.pinhole_intrinsics(H, W)
    N = 512
    pix = (synth.hash_u01(N, 99) * (H * W)).astype(np.int64)
    res = ru.get_rays(torch.from_numpy(pose)[None], np.array([fx, fy, cx, cy], dtype=np.float32), H, W, -1)
    ro = res["rays_o"].reshape(-1, 3)[pix].contiguous(); rd = res["rays_d"].reshape(-1, 3)[pix].contiguous()
    labels = torch.from_numpy((synth.hash_u01(N, 100) < 0.5).astype(np.int64))
    out = model.render(ro, rd, staged=False, bg_color=1, perturb=False, update_proposal=False,
                       return_rgb=0, return_feats=0, return_mask=1)
    logits = out["instance_mask_logits"]
    pm = torch.softmax(logits, dim=-1).clamp(min=opt.epsilon, max=1 - opt.epsilon)
    loss = (-torch.log(torch.gather(pm, -1, labels[..., None]))).mean()
    loss.backward()
    g_emb = model.m_grid.embeddings.grad
    touched = torch.nonzero(g_emb.abs().sum(-1) > 0).squeeze(-1)
    # fixture keeps a deterministic 4096-row sample of the touched rows plus whole-table checksums
    pick = np.unique((synth.hash_u01(4096, 7) * touched.numel()).astype(np.int64))
    rows = touched[torch.from_numpy(pick)]
    sav = dict(rays_o=np_(ro), rays_d=np_(rd), labels=np_(labels), pixels=pix, pose=pose,
               logits=np_(logits), loss=np.array(loss.item()), param_spec=np.array(json.dumps(spec)),
               m_grid_rows=np_(rows).astype(np.int64), m_grid_grad_rows=np_(g_emb[rows]),
               m_grid_touched=np.array(touched.numel()), m_grid_grad_sum=np.array(g_emb.double().sum().item()),
               m_grid_grad_abssum=np.array(g_emb.double().abs().sum().item()),
               epsilon=np.array(opt.epsilon))
    for i, l in enumerate(model.mask_mlp[0].net):
        sav[f"mask_mlp_grad{i}"] = np_(l.weight.grad)
    print(f"   loss {loss.item():.6f}; touched m_grid rows {touched.numel()}; |grad|max {g_emb.abs().max().item():.3e}")
    np.savez_compressed(os.path.join(GOLD, "train_c5.npz"), **sav)


def fx_train_rgb(rr, nn_, ru):
    """RGB-mode training step (trainer.py:360-392): everything trainable, MSE(image, gt) + lambda_proposal *
    proposal_loss (renderer.py:30-57); the distortion term needs the absent third-party torch_efficient_distloss
    and is switched off.  perturb=False so that the step is deterministic.  256 rays."""
    print("[train:rgb] radiance-field MSE + proposal-loss step, 256 rays")
    opt = make_opt(num_steps=[128, 64, 32], with_sam=False, with_mask=False)
    opt.lambda_proposal, opt.lambda_distort = 1.0, 0.0
    torch.manual_seed(0)
    model = nn_.NeRFNetwork(opt)
    spec = param_spec_for(model, 777, 1.0, 4.0)
    load_params(model, spec)
    model.train()
    H = W = 256
    pose = synth.orbit_pose(1.2, 15.0, 100.0)
    fx, fy, cx, cy = synth.pinhole_intrinsics(H, W)
    N = 256
    pix = (synth.hash_u01(N, 41) * (H * W)).astype(np.int64)
    res = ru.get_rays(torch.from_numpy(pose)[None], np.array([fx, fy, cx, cy], dtype=np.float32), H, W, -1)
    ro = res["rays_o"].reshape(-1, 3)[pix].contiguous(); rd = res["rays_d"].reshape(-1, 3)[pix].contiguous()
    gt = torch.from_numpy(synth.hash_uniform((N, 3), 42, 0.0, 1.0))
    out = model.render(ro, rd, staged=False, bg_color=1, perturb=False, update_proposal=True)
    mse = torch.nn.MSELoss(reduction="none")(out["image"], gt).mean()
    loss = mse + opt.lambda_proposal * out["proposal_loss"]
    loss.backward()
    sav = dict(rays_o=np_(ro), rays_d=np_(rd), gt=np_(gt), pixels=pix, pose=pose, image=np_(out["image"]),
               proposal_loss=np.array(out["proposal_loss"].item()), mse=np.array(mse.item()), loss=np.array(loss.item()),
               param_spec=np.array(json.dumps(spec)))
    for name, p in model.named_parameters():
        g = p.grad
        assert g is not None, name
        if name.endswith("embeddings"):      # tables: a deterministic sample of touched rows + whole-table checksums
            touched = torch.nonzero(g.abs().sum(-1) > 0).squeeze(-1)
            pick = np.unique((synth.hash_u01(2048, 11) * touched.numel()).astype(np.int64))
            rows = touched[torch.from_numpy(pick)]
            sav[f"rows:{name}"] = np_(rows).astype(np.int64)
            sav[f"grad_rows:{name}"] = np_(g[rows])
            sav[f"touched:{name}"] = np.array(touched.numel())
            sav[f"abssum:{name}"] = np.array(g.double().abs().sum().item())
            sav[f"sum:{name}"] = np.array(g.double().sum().item())
        else:
            sav[f"grad:{name}"] = np_(g)
    print(f"   mse {mse.item():.6f} proposal {out['proposal_loss'].item():.6f}")
    np.savez_compressed(os.path.join(GOLD, "train_rgb.npz"), **sav)


def fx_train_sam(rr, nn_, ru):
    """SAM-feature distillation step (trainer.py:505-549, cache branch): radiance field frozen (main.py:249-256),
    s_grid + samvit_mlp trainable; low-res feature render (return_feats=1, perturb=False) -> [1,256,h,w] ->
    F.interpolate(bilinear) to the SAM feature map size -> MSE.  24x24 rays here (64x64 in the reference)."""
    import torch.nn.functional as F
    print("[train:sam] SAM-feature distillation step, 24x24 rays")
    opt = make_opt(num_steps=[128, 64, 32], with_sam=True, with_mask=False)
    torch.manual_seed(0)
    model = nn_.NeRFNetwork(opt)
    spec = param_spec_for(model, 31337, 1.0, 4.0)
    load_params(model, spec)
    for n_, p in model.named_parameters():
        p.requires_grad_(n_.startswith("s_grid") or n_.startswith("samvit_mlp"))
    model.train()
    h = w = 24
    pose = synth.orbit_pose(1.0, 30.0, 200.0)
    fx, fy, cx, cy = synth.pinhole_intrinsics(h, w)
    res = ru.get_rays(torch.from_numpy(pose)[None], np.array([fx, fy, cx, cy], dtype=np.float32), h, w, -1)
    ro = res["rays_o"].reshape(-1, 3).contiguous(); rd = res["rays_d"].reshape(-1, 3).contiguous()
    gt = torch.from_numpy(synth.hash_uniform((1, 256, 32, 32), 77, -1.0, 1.0))      # "SAM" map of another size: the resize is exercised
    out = model.render(ro, rd, staged=False, bg_color=1, perturb=False, return_feats=1, H=h, W=w)
    pred = out["samvit"].reshape(1, h, w, 256).permute(0, 3, 1, 2).contiguous()
    pred = F.interpolate(pred, gt.shape[2:], mode="bilinear")
    loss = torch.nn.MSELoss(reduction="none")(pred, gt).mean()
    loss.backward()
    # gt is regenerated by the test from its seed: synth.hash_uniform((1, 256, 32, 32), 77, -1, 1)
    sav = dict(rays_o=np_(ro), rays_d=np_(rd), gt_seed=np.array(77), gt_shape=np.array([1, 256, 32, 32]), pose=pose,
               h=np.array(h), w=np.array(w), samvit=np_(out["samvit"]),
               loss=np.array(loss.item()), param_spec=np.array(json.dumps(spec)))
    g = model.s_grid.embeddings.grad
    touched = torch.nonzero(g.abs().sum(-1) > 0).squeeze(-1)
    pick = np.unique((synth.hash_u01(2048, 13) * touched.numel()).astype(np.int64))
    rows = touched[torch.from_numpy(pick)]
    sav.update({"s_grid_rows": np_(rows).astype(np.int64), "s_grid_grad_rows": np_(g[rows]), "s_grid_touched": np.array(touched.numel()),
                "s_grid_grad_abssum": np.array(g.double().abs().sum().item())})
    for name, p in model.named_parameters():
        if name.startswith("samvit_mlp"):
            g_ = np_(p.grad).reshape(-1)
            if g_.size > 20000:      # big layers: every 11th entry + the norm (keeps the fixture small)
                sav[f"grad11:{name}"] = g_[::11].copy()
                sav[f"gradnorm:{name}"] = np.array(np.linalg.norm(g_.astype(np.float64)))
            else:
                sav[f"grad:{name}"] = np_(p.grad)
        elif not name.startswith("s_grid"):
            assert p.grad is None, name
    print(f"   loss {loss.item():.6f}; touched s_grid rows {touched.numel()}")
    np.savez_compressed(os.path.join(GOLD, "train_sam.npz"), **sav)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    rr, nn_, ru, enc = install_reference()
    want = set(args.only.split(",")) if args.only else None

    def on(n):
        return want is None or n in want
    if on("units"):
        fx_units(rr, ru, enc)
    if on("sref"):
        fx_render(rr, nn_, ru, "sref", [128, 64, 32], False, 16, 64, 64, 1234, 1.0, 4.0, time_it=True)
    if on("flat128"):
        fx_render(rr, nn_, ru, "flat128", [128], False, 12, 64, 64, 555, 1.0, 4.0)
    if on("flat128_h"):     # the bench configuration's storage: tables rounded to fp16 values (BASELINE configs[1])
        fx_render(rr, nn_, ru, "flat128_h", [128], False, 12, 64, 64, 777, 1.0, 4.0, tables_f16=True)
    if on("sref_h"):
        fx_render(rr, nn_, ru, "sref_h", [128, 64, 32], False, 12, 64, 64, 4321, 1.0, 4.0, tables_f16=True)
    if on("heads"):
        fx_render(rr, nn_, ru, "heads", [128, 64, 32], True, 8, 64, 64, 999, 1.0, 4.0)
    if on("c1"):
        fx_c1(rr, nn_, enc)
    if on("train"):
        fx_train(rr, nn_, ru)
    if on("train_rgb"):
        fx_train_rgb(rr, nn_, ru)
    if on("train_sam"):
        fx_train_sam(rr, nn_, ru)


if __name__ == "__main__":
    main()
