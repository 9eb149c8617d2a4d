"""CPU: the C-ABI library loads and exports every symbol include/sanerf_hip.h declares; the host
logic (layouts, state_dict keys, sharding, drop-in names) behaves like the reference's.  No
compute calls (there is no GPU here)."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from helpers import ROOT, golden, make_opt, spec_of


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "sanerf_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(sn_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from sanerf_hq_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "libsanerf_hip.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sanerf_hip.h but not exported"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names, "ctypes signature table out of sync with the header"
    assert lib.sn_abi_version() == _lib.ABI_VERSION == 12
    assert lib.sn_build_flags() == 0, "the product library carries neither the experiment kernels nor the LDS poisoning"


def test_round6_entry_points_validate_their_arguments():
    """The training entry points added with ABI 11 answer bad arguments with error codes before any launch."""
    from sanerf_hq_amd import _lib
    l = _lib.lib()
    d = _lib.MlpDesc()
    d.num_layers, d.activation, d.skip_mask = 3, 0, 0
    for i, v in enumerate((32, 64, 64, 16)):
        d.dims[i] = v
    assert l.sn_mlp_small_supported(ctypes.byref(d)) == 1
    d.dims[1] = 48
    assert l.sn_mlp_small_supported(ctypes.byref(d)) == 0
    dummy = ctypes.c_void_p(16)
    for i in range(3):
        d.weight[i] = 16
    hid = (ctypes.c_void_p * 2)(16, 16)
    assert l.sn_mlp_small_forward_train(ctypes.byref(d), dummy, 4, hid, dummy, 0, None, 0.0, None, None) == -2      # widths not instantiated
    assert b"not instantiated" in l.sn_last_error()
    d.dims[1] = 64
    assert l.sn_mlp_small_forward_train(ctypes.byref(d), dummy, 4, hid, dummy, 1, None, 0.0, None, None) == -1      # activated output without destination
    assert l.sn_mlp_small_forward_train(ctypes.byref(d), dummy, 4, hid, dummy, 7, None, 0.0, dummy, None) == -1
    assert b"unknown output activation" in l.sn_last_error()
    assert l.sn_mlp_small_forward_train(ctypes.byref(d), None, 0, hid, None, 0, None, 0.0, None, None) == 0           # an empty batch launches nothing
    assert l.sn_mlp_small_backward(ctypes.byref(d), None, None, 0, None, 0.0, hid, 4, None, hid, dummy, None, None) == -1
    assert b"no incoming gradient" in l.sn_last_error()
    d.bias[0] = 16
    assert l.sn_mlp_small_supported(ctypes.byref(d)) == 0
    assert l.sn_rm_jitter(None, 4, 8, 2, dummy, None) == -1 and b"kind" in l.sn_last_error()
    assert l.sn_rm_jitter(None, 0, 8, 0, None, None) == 0 and l.sn_rm_ray_composite(None, None, None, None, 0, 8, None, None, None, None) == 0
    assert l.sn_zero(ctypes.c_void_p(8), 16, None) == -1
    assert l.sn_rm_sample_positions_ex(dummy, dummy, dummy, dummy, dummy, 4, 8, 1, -1.0, dummy, dummy, dummy, None) == -1
    assert l.sn_rm_ray_composite(None, dummy, dummy, dummy, 4, 8, dummy, dummy, dummy, None) == -1


def test_abi12_entry_points_validate_their_arguments():
    """sn_gemm_f32, sn_rm_proposal_loss_long and the length checks of the long-ray kernels answer bad arguments with error codes before any launch."""
    from sanerf_hq_amd import _lib
    l = _lib.lib()
    dummy = ctypes.c_void_p(16)
    assert l.sn_gemm_f32(dummy, 3, 2, dummy, 1, 8, None, 0, 4, 4, 8, dummy, 4, None) == -1 and b"one stride" in l.sn_last_error()      # neither stride of a is 1
    assert l.sn_gemm_f32(dummy, 8, 1, dummy, 1, 8, None, 3, 4, 4, 8, dummy, 4, None) == -1 and b"activation" in l.sn_last_error()
    assert l.sn_gemm_f32(dummy, 8, 1, dummy, 1, 8, None, 0, 4, 4, 8, dummy, 3, None) == -1 and b"row stride" in l.sn_last_error()
    assert l.sn_gemm_f32(None, 8, 1, dummy, 1, 8, None, 0, 4, 4, 8, dummy, 4, None) == -1
    assert l.sn_gemm_f32(None, 8, 1, None, 1, 8, None, 0, 0, 4, 8, None, 4, None) == 0                                                  # no rows: nothing to launch
    # proposal loss: up to 512 samples per ray no workspace, beyond that the size the library states (at most 64 MiB)
    assert l.sn_rm_proposal_loss_workspace_bytes(4096, 128, 32, 1) == 0 and l.sn_rm_proposal_loss_workspace_bytes(0, 4000, 4000, 1) == 0
    need = l.sn_rm_proposal_loss_workspace_bytes(4096, 600, 700, 1)
    assert 0 < need <= (64 << 20) and need % 8 == 0
    assert l.sn_rm_proposal_loss_workspace_bytes(4, 600, 700, 0) < l.sn_rm_proposal_loss_workspace_bytes(4, 600, 700, 1) < need
    assert l.sn_rm_proposal_loss_long(dummy, dummy, dummy, dummy, 4096, 600, 700, 1.0, None, None, dummy, None, 0, None) == -1 and b"workspace" in l.sn_last_error()
    assert l.sn_rm_proposal_loss_long(dummy, dummy, dummy, dummy, 4096, 600, 700, 1.0, None, None, dummy, ctypes.c_void_p(12), need, None) == -1      # misaligned
    assert l.sn_rm_proposal_loss_long(dummy, dummy, dummy, dummy, 4096, 600, 700, 1.0, None, dummy, dummy, dummy, need, None) == -1                    # both outputs
    assert l.sn_rm_proposal_loss_scaled(dummy, dummy, dummy, dummy, 4096, 600, 700, 1.0, None, None, dummy, None) == -1 and b"sn_rm_proposal_loss_long" in l.sn_last_error()
    assert l.sn_rm_proposal_loss_long(dummy, dummy, dummy, dummy, 0, 600, 700, 1.0, None, None, dummy, None, 0, None) == 0
    assert l.sn_rm_weights_from_sigma_backward(dummy, dummy, dummy, 4, 200000, 1, dummy, None) == -1 and b"131072" in l.sn_last_error()
    assert l.sn_rm_distort_loss(dummy, dummy, 4, 0, dummy, dummy, None) == -1


def test_library_reports_missing_device_loudly():
    from sanerf_hq_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    rc = _lib.lib().sn_device_count()
    assert rc < 0 and b"HIP error" in _lib.lib().sn_last_error()


def test_invalid_arguments_return_error_codes_not_exceptions():
    from sanerf_hq_amd import _lib
    l = _lib.lib()
    offs = _lib.host_i32([0, 8, 16])
    # NULL pointers
    assert l.sn_grid_encode_forward(None, None, 0, offs, None, 4, 3, 2, 2, 2, 0.5, 16, None, 0, 0, 0, 1, None) == -1
    assert b"device pointers" in l.sn_last_error()
    # the reference's own messages for unsupported D / C (gridencoder.cu:392,409), checked before any launch
    dummy = ctypes.c_void_p(16)
    assert l.sn_grid_encode_forward(dummy, dummy, 0, offs, dummy, 4, 7, 2, 2, 2, 0.5, 16, None, 0, 0, 0, 1, None) == -1
    assert b"D must be 2, 3, 4 or 5" in l.sn_last_error()
    assert l.sn_grid_encode_forward(dummy, dummy, 0, offs, dummy, 4, 3, 3, 2, 2, 0.5, 16, None, 0, 0, 0, 1, None) == -1
    assert b"C must be 1, 2, 4, 8, 16 or 32" in l.sn_last_error()
    assert l.sn_sh_encode_forward(dummy, dummy, 4, 3, 9, None, None) == -1
    assert b"degree in [1, 8]" in l.sn_last_error()
    assert l.sn_freq_encode_forward(dummy, 4, 3, 4, 26, dummy, None) == -1


def test_python_operators_refuse_cpu_tensors():
    """No silent CPU path: the operators raise like the reference's CHECK_CUDA (gridencoder.cu:15)."""
    from sanerf_hq_amd.gridencoder import GridEncoder
    from sanerf_hq_amd.shencoder import SHEncoder
    from sanerf_hq_amd.freqencoder import FreqEncoder
    from sanerf_hq_amd import raymarching as rm
    enc = GridEncoder(num_levels=2, log2_hashmap_size=8, desired_resolution=32)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        enc(torch.rand(4, 3))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        SHEncoder()(torch.rand(4, 3))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        FreqEncoder()(torch.rand(4, 3))
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        rm.contract(torch.rand(4, 3))


def test_grid_encoder_layout_matches_reference_recipe(orc):
    from sanerf_hq_amd.gridencoder import GridEncoder
    for kw in (dict(num_levels=16, level_dim=2, log2_hashmap_size=19, desired_resolution=4096),
               dict(num_levels=16, level_dim=8, log2_hashmap_size=19, desired_resolution=512),
               dict(num_levels=5, level_dim=2, log2_hashmap_size=17, desired_resolution=128),
               dict(num_levels=8, level_dim=2, log2_hashmap_size=14, desired_resolution=2048),
               dict(num_levels=16, level_dim=2, log2_hashmap_size=10, desired_resolution=256)):
        enc = GridEncoder(input_dim=3, **kw)
        offs, pls = orc.grid_layout(3, kw["num_levels"], kw["level_dim"], 2, 16, kw["log2_hashmap_size"], kw["desired_resolution"])
        assert np.array_equal(enc.offsets.numpy(), offs) and enc.offsets.dtype == torch.int32
        assert enc.per_level_scale == pls and enc.output_dim == kw["num_levels"] * kw["level_dim"]
        assert tuple(enc.embeddings.shape) == (int(offs[-1]), kw["level_dim"])
        assert float(enc.embeddings.detach().abs().max()) <= 1e-4          # grid.py:144-146 init
    assert set(GridEncoder(num_levels=2, log2_hashmap_size=8).state_dict()) == {"embeddings", "offsets"}


def test_network_state_dict_matches_reference_keys_and_shapes():
    """param_spec in the fixture was read off the reference's own NeRFNetwork.state_dict()."""
    from sanerf_hq_amd.nerf import NeRFNetwork
    spec = spec_of(golden("render_heads"))
    model = NeRFNetwork(make_opt(with_sam=True, with_mask=True))
    sd = model.state_dict()
    ref = {s["name"]: tuple(s["shape"]) for s in spec}
    ours = {k: tuple(v.shape) for k, v in sd.items() if not (k.endswith("offsets") or k.startswith("aabb"))}
    assert ours == ref
    assert {"aabb_train", "aabb_infer", "grid.offsets", "prop_encoders.1.offsets", "s_grid.offsets", "m_grid.offsets"} <= set(sd)
    groups = model.get_params(1e-2)
    assert len(groups) == 9 and all(g["lr"] == 1e-2 for g in groups)


def test_encoder_factory_names():
    from sanerf_hq_amd.encoding import get_encoder
    enc, dim = get_encoder("hashgrid", num_levels=4, log2_hashmap_size=8, desired_resolution=64)
    assert dim == 8 and enc.gridtype == "hash"
    enc, dim = get_encoder("tiledgrid", num_levels=4, log2_hashmap_size=8, desired_resolution=64)
    assert enc.gridtype == "tiled"
    assert get_encoder("sh", degree=4)[1] == 16
    assert get_encoder("frequency", multires=6)[1] == 39
    fn, dim = get_encoder("None")
    assert dim == 3 and fn(5) == 5
    t, dim = get_encoder("frequency_torch", multires=4)
    x = torch.rand(7, 3)
    assert dim == 27 and torch.allclose(t(x)[:, 3:6], torch.sin(x))
    with pytest.raises(NotImplementedError):
        get_encoder("bogus")


def test_dropin_module_names():
    import sys
    import sanerf_hq_amd
    sanerf_hq_amd.install_dropin()
    import activation, encoding, freqencoder, gridencoder, raymarching, shencoder   # noqa: F401
    assert gridencoder.GridEncoder is sys.modules["sanerf_hq_amd.gridencoder"].GridEncoder
    assert callable(gridencoder.grid_encode) and callable(shencoder.sh_encode) and callable(freqencoder.freq_encode)
    assert callable(activation.trunc_exp) and callable(encoding.get_encoder)
    for name in ("generate_rays", "near_far_from_aabb", "contract", "sample_pdf", "weights_from_sigma", "composite", "render_rays"):
        assert callable(getattr(raymarching, name))


def test_trunc_exp_backward_clamps():
    from sanerf_hq_amd.activation import trunc_exp
    x = torch.tensor([-20.0, 0.0, 3.0, 20.0], requires_grad=True)
    y = trunc_exp(x)
    y.sum().backward()
    assert torch.allclose(y, torch.exp(x.detach()))
    assert torch.allclose(x.grad, torch.exp(x.detach().clamp(-15, 15)))


def test_shard_rows_cover_image_in_tile_aligned_bands():
    from sanerf_hq_amd.dist import all_shards
    for H in (16, 64, 800, 1600, 1000, 17):
        for world in (1, 2, 4, 8):
            bands = all_shards(H, world)
            assert bands[0][0] == 0 and bands[-1][1] == H
            for (b0, e0), (b1, e1) in zip(bands, bands[1:]):
                assert e0 == b1 and (b0 % 16 == 0 or b0 == e0)
            sizes = [e - b for b, e in bands]
            if H >= 16 * world:
                assert max(sizes) - min(sizes) <= 16 + (16 - H % 16) % 16   # balanced to one tile row


def test_synth_is_bit_deterministic():
    from sanerf_hq_amd import synth
    a = synth.hash_uniform((5,), 1234, -1, 1)
    assert a.dtype == np.float32
    assert a.view(np.uint32).tolist() == [3204617886, 3210456146, 3194325912, 3209874940, 3204665038]   # same bits on every machine
    assert np.array_equal(synth.hash_uniform((1000,), 9, 0, 1), synth.hash_uniform((1000,), 9, 0, 1))
    u = synth.hash_u01(100000, 3)
    assert 0.49 < u.mean() < 0.51 and u.min() >= 0 and u.max() < 1
    p = synth.orbit_pose(1.0, 20.0, 30.0)
    R = p[:3, :3].astype(np.float64)
    np.testing.assert_allclose(R.T @ R, np.eye(3), atol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(p[:3, 3]), 1.0, atol=1e-6)
    np.testing.assert_allclose(-p[:3, 2], -p[:3, 3] / np.linalg.norm(p[:3, 3]), atol=1e-6)   # looks at the origin


def test_losses_on_cpu():
    """proposal / distortion losses are torch-only helpers of the differentiable path."""
    from sanerf_hq_amd.nerf.renderer import distort_loss, proposal_loss
    torch.manual_seed(0)
    bins = torch.sort(torch.rand(6, 9), -1).values
    w = torch.rand(6, 8); w = w / w.sum(-1, keepdim=True)
    m = bins[:, :-1] + (bins[:, 1:] - bins[:, :-1]) / 2
    d = bins[:, 1:] - bins[:, :-1]
    brute = (w[:, :, None] * w[:, None, :] * (m[:, :, None] - m[:, None, :]).abs()).sum((-1, -2)) + (w * w * d).sum(-1) / 3
    assert torch.allclose(distort_loss(bins, w), brute.mean(), atol=1e-6)
    # a proposal histogram that upper-bounds the reference one has zero loss
    assert float(proposal_loss([bins, bins], [w * 2, w])) == 0.0
    assert float(proposal_loss([bins, bins], [w * 0.5, w])) > 0.0


def test_checkpoint_roundtrip_in_reference_format(tmp_path):
    """nerf/trainer.py:1685-1741 / :1779-1800: {'epoch','global_step','stats','model'}; a pretrained radiance field
    loads into a SAM/mask model with the heads missing (strict=False) and is frozen like main.py:249-256."""
    from sanerf_hq_amd.nerf import NeRFNetwork
    from sanerf_hq_amd.nerf.utils import freeze_loaded_parameters, load_checkpoint, save_checkpoint
    torch.manual_seed(3)
    base = NeRFNetwork(make_opt())
    with torch.no_grad():
        for p in base.parameters():
            p.uniform_(-0.5, 0.5)
    path = str(tmp_path / "ngp_ep0007.pth")
    state = save_checkpoint(base, path, epoch=7, global_step=1234)
    assert set(state) == {"epoch", "global_step", "stats", "model"} and set(state["stats"]) >= {"loss", "valid_loss", "results", "checkpoints", "best_result"}
    fresh = NeRFNetwork(make_opt())
    missing, unexpected, st = load_checkpoint(fresh, path, map_location="cpu")
    assert not missing and not unexpected and st["epoch"] == 7 and st["global_step"] == 1234
    for (k, a), (_, b) in zip(base.state_dict().items(), fresh.state_dict().items()):
        assert torch.equal(a, b), k
    heads = NeRFNetwork(make_opt(with_sam=True, with_mask=True))
    missing, unexpected, st = load_checkpoint(heads, path, map_location="cpu")
    assert not unexpected and all(m.split(".")[0] in ("s_grid", "samvit_mlp", "m_grid", "mask_mlp") for m in missing)
    frozen = freeze_loaded_parameters(heads, st["model"])
    assert "grid.embeddings" in frozen and all(not heads.get_parameter(k).requires_grad for k in frozen)
    assert heads.s_grid.embeddings.requires_grad and heads.mask_mlp[0].net[0].weight.requires_grad
    bare = NeRFNetwork(make_opt())
    assert load_checkpoint(bare, base.state_dict()) == ([], [], {})          # bare state_dict branch (trainer.py:1793-1796)


def test_checkpoint_layout_equals_the_reference(tmp_path):
    """tests/golden/checkpoint_layout.json was captured from the REFERENCE's NeRFNetwork (tools/gen_ckpt_fixture.py): ordered
    state_dict keys / shapes / dtypes, Adam param-group structure of get_params, top-level checkpoint keys and the stats
    dictionary of trainer.py:151-157, 1685-1718 -- for every training mode of main.py."""
    import json
    from sanerf_hq_amd.nerf import NeRFNetwork, load_checkpoint, save_checkpoint
    lay = json.load(open(os.path.join(ROOT, "tests", "golden", "checkpoint_layout.json")))
    for mode, kw in (("rgb", {}), ("sam", dict(with_sam=True)), ("mask", dict(with_mask=True)), ("sam+mask", dict(with_sam=True, with_mask=True))):
        ref = lay["modes"][mode]
        model = NeRFNetwork(make_opt(**kw))
        sd = model.state_dict()
        ours = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]
        assert ours == ref["state_dict"], f"{mode}: state_dict differs from the reference's (order, shapes or dtypes)"
        for k, v in ref["small_buffers"].items():
            assert sd[k].tolist() == v, f"{mode}: buffer {k}"
        groups = model.get_params(1e-2)
        optim = torch.optim.Adam([dict(params=list(g["params"]), lr=g["lr"]) for g in groups], betas=(0.9, 0.99), eps=1e-15)
        assert [len(g["params"]) for g in optim.state_dict()["param_groups"]] == ref["param_groups"]
        if mode != "mask":
            continue
        # a checkpoint as the reference writes it (full=True) loads into the product and round-trips through the writer
        sched = torch.optim.lr_scheduler.LambdaLR(optim, lambda it: 1.0)
        path = os.path.join(tmp_path, "ngp_ep0003.pth")
        state = save_checkpoint(model, path, epoch=3, global_step=77, optimizer=optim, lr_scheduler=sched)
        assert set(lay["top_level_keys_default"]) <= set(state) and set(state) <= set(lay["top_level_keys_full"])
        assert list(state["stats"]) == list(lay["stats"]) and state["stats"] == lay["stats"]
        fresh = NeRFNetwork(make_opt(**kw))
        missing, unexpected, loaded = load_checkpoint(fresh, path, map_location="cpu")
        assert missing == [] and unexpected == [] and loaded["epoch"] == 3 and loaded["global_step"] == 77
        assert torch.equal(fresh.mask_mlp[0].net[0].weight, model.mask_mlp[0].net[0].weight)
        # an RGB-mode reference checkpoint into the mask model: only the new heads are missing (main.py:243-256)
        rgb_sd = {k: torch.zeros(shape, dtype=getattr(torch, dt)) for k, shape, dt in lay["modes"]["rgb"]["state_dict"]}
        missing, unexpected, _ = load_checkpoint(fresh, {"model": rgb_sd, "epoch": 1, "global_step": 1, "stats": lay["stats"]})
        assert unexpected == [] and sorted(missing) == sorted(k for k, _, _ in ref["state_dict"] if k.startswith(("m_grid", "mask_mlp")))


def test_sam_feature_cache_container(tmp_path):
    """<workspace>/sam_cache/<img_name>.npy, float32 [256,64,64] (trainer.py:1069-1079), read back as [1,256,64,64] (:924-926)."""
    from sanerf_hq_amd.nerf import SamFeatureCache, feature_map
    cache = SamFeatureCache(str(tmp_path))
    f = torch.randn(1, 256, 64, 64)
    path = cache.store("frame_00012", f)
    assert path == os.path.join(str(tmp_path), "sam_cache", "frame_00012.npy") and "frame_00012" in cache and "nope" not in cache
    raw = np.load(path)
    assert raw.dtype == np.float32 and raw.shape == (256, 64, 64)              # what the reference's np.save wrote / np.load expects
    back = cache.load("frame_00012")
    assert back.shape == (1, 256, 64, 64) and torch.equal(back, f)
    np.save(os.path.join(cache.path, "bad.npy"), np.zeros((64, 64), np.float64))
    with pytest.raises(ValueError, match="float32"):
        cache.load("bad")
    # rendered per-ray features -> the map the reference compares with the cached target (trainer.py:536-542)
    samvit = torch.randn(32 * 32, 256)
    m = feature_map(samvit, 32, 32, size=(64, 64))
    want = torch.nn.functional.interpolate(samvit.reshape(1, 32, 32, 256).permute(0, 3, 1, 2), (64, 64), mode="bilinear")
    assert m.shape == (1, 256, 64, 64) and torch.equal(m, want)


def test_round2_host_logic_has_no_cpu_fallback():
    """The Python mirror of the round-2 entry points refuses CPU tensors instead of falling back (the product path must
    fail loudly without the HIP kernels), and the mask head's fusability rule is plain host logic."""
    import torch
    from sanerf_hq_amd import raymarching as rm
    from sanerf_hq_amd.gridencoder import GridEncoder
    from sanerf_hq_amd.nerf.network import SkipConnMLP
    from sanerf_hq_amd.optim import Adam
    p = torch.nn.Parameter(torch.zeros(8))
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="CUDA"):
        Adam([p], lr=1e-3).step()
    with pytest.raises(ValueError):
        Adam([p], lr=1e-3, amsgrad=True)
    enc = GridEncoder(input_dim=3, num_levels=16, level_dim=8, base_resolution=16, log2_hashmap_size=10, desired_resolution=64)
    mlp = SkipConnMLP(143, 2, 256, 3, skip_layers=[], bias=False)
    assert rm.mask_head_fusable(enc, mlp, 32, 15)
    assert not rm.mask_head_fusable(enc, mlp, 48, 15)                                   # samples per ray not a power of two
    assert not rm.mask_head_fusable(enc, mlp, 256, 15)                                  # more than 128 samples per ray
    assert not rm.mask_head_fusable(enc, SkipConnMLP(143, 2, 256, 4, skip_layers=[2], bias=False), 32, 15)   # skip layer
    assert not rm.mask_head_fusable(enc, SkipConnMLP(143, 2, 128, 3, skip_layers=[], bias=False), 32, 15)    # hidden width
    assert not rm.mask_head_fusable(enc, mlp, 32, 14)                                   # input width does not match
    with pytest.raises(RuntimeError, match="CUDA"):
        rm.mask_head(torch.zeros(2, 32), torch.zeros(2, 32, 3), torch.zeros(2, 32, 15), enc, mlp, 2.0)


def test_fused_route_selection_is_host_logic():
    """NeRFRenderer._fused_kind(): which last-stage kernel (if any) sn_rm_render_rays has for a field -- the reference network's own sizes
    ("main"), another field of the same structure within the size-agnostic kernel's limits ("any": BASELINE configs[0]), or neither (the
    operator chain: a non-standard forward(), biases, layers wider than 64, another level_dim)."""
    from sanerf_hq_amd.encoding import get_encoder
    from sanerf_hq_amd.nerf import NeRFNetwork
    from sanerf_hq_amd.nerf.network import MLP
    from sanerf_hq_amd.nerf.renderer import NeRFRenderer
    from sanerf_hq_amd.synth import make_opt
    assert NeRFNetwork(make_opt(num_steps=[128, 64, 32]))._fused_kind() == "main"

    class Small(NeRFRenderer):
        def __init__(self, opt, level_dim=2, hidden=32, bias=False):
            super().__init__(opt)
            self.grid, d = get_encoder("hashgrid", input_dim=3, level_dim=level_dim, num_levels=8, log2_hashmap_size=14, desired_resolution=2048)
            self.grid_mlp = MLP(d, 16, hidden, 2, bias=bias)
            self.view_encoder, vd = get_encoder("sh", input_dim=3, degree=4)
            self.view_mlp = MLP(15 + vd, 3, 32, 2, bias=False)

    m = Small(make_opt(num_steps=[32]))
    assert m._fused_kind() == "any" and m._fused_shape() and m.fused_min_rays == 16384
    m.standard_field = False
    assert m._fused_kind() is None and not m._fused_shape()
    assert Small(make_opt(num_steps=[32]), hidden=96)._fused_kind() is None           # wider than the kernel's 64
    assert Small(make_opt(num_steps=[32]), bias=True)._fused_kind() is None           # biases: not the reference's structure
    assert Small(make_opt(num_steps=[32]), level_dim=4)._fused_kind() is None         # the size-agnostic kernel is built for level_dim 2
    assert Small(make_opt(num_steps=[32, 16]))._fused_kind() is None                  # a proposal stage without the reference's proposal networks


@pytest.mark.parametrize("T_", [32, 64, 128])
def test_distort_loss_statement_vs_fp64_brute_force_with_degenerate_rays(T_):
    """nerf.renderer.distort_loss (the O(T) cumulative-sum statement; what the reference gets from the third-party eff_distloss,
    renderer.py:17-27) against the published O(T^2) definition in fp64, value and gradient, on random rays, zero-width intervals, all mass in
    one bin and an empty ray.  The HIP kernel is held to the same brute force in tests/test_gpu_ops.py."""
    from sanerf_hq_amd.nerf import renderer as R
    rng = np.random.default_rng(T_)
    N = 16
    b = np.sort(rng.uniform(0, 1, (N, T_ + 1)), axis=1)
    w = rng.uniform(0, 1, (N, T_)) ** 4
    b[1, T_ // 3: 2 * T_ // 3 + 1] = b[1, T_ // 3]
    w[2] = 0.0; w[2, T_ // 2] = 1.0
    w[3] = 0.0
    b[4] = 0.5
    bd = torch.from_numpy(b)
    wd = torch.from_numpy(w).requires_grad_(True)
    d = bd[:, 1:] - bd[:, :-1]
    m = bd[:, :-1] + d / 2
    ref = ((wd[:, :, None] * wd[:, None, :] * (m[:, :, None] - m[:, None, :]).abs()).sum((1, 2)) + (wd * wd * d).sum(1) / 3).mean()
    ref.backward()
    w2 = torch.from_numpy(w).requires_grad_(True)
    got = R.distort_loss(bd, w2)
    got.backward()
    assert abs(float(got.detach()) - float(ref.detach())) <= 1e-12 * max(1.0, abs(float(ref.detach())))
    assert float((w2.grad - wd.grad).abs().max()) <= 1e-12 * max(1.0, float(wd.grad.abs().max()))
