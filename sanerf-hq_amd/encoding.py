"""Encoder factory with the reference's names and return convention (encoding.py:47-79):
`get_encoder(encoding, ...) -> (module, output_dim)`."""
import torch
import torch.nn as nn


class FreqEncoder_torch(nn.Module):
    """Pure-torch positional encoding kept for the 'frequency_torch' key (reference encoding.py:6-44);
    same call signature and output ordering."""

    def __init__(self, input_dim, max_freq_log2, N_freqs, log_sampling=True, include_input=True,
                 periodic_fns=(torch.sin, torch.cos)):
        super().__init__()
        self.input_dim = input_dim
        self.include_input = include_input
        self.periodic_fns = periodic_fns
        self.output_dim = (input_dim if include_input else 0) + input_dim * N_freqs * len(periodic_fns)
        bands = 2.0 ** torch.linspace(0.0, max_freq_log2, N_freqs) if log_sampling \
            else torch.linspace(2.0 ** 0.0, 2.0 ** max_freq_log2, N_freqs)
        self.freq_bands = bands.numpy().tolist()

    def forward(self, input, **kwargs):
        parts = [input] if self.include_input else []
        for freq in self.freq_bands:
            parts.extend(fn(input * freq) for fn in self.periodic_fns)
        return torch.cat(parts, dim=-1)


def get_encoder(encoding, input_dim=3, multires=6, degree=4, num_levels=16, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=2048, align_corners=False, interpolation="linear", **kwargs):
    if encoding == "None":
        return (lambda x, **kw: x), input_dim
    if encoding == "frequency_torch":
        encoder = FreqEncoder_torch(input_dim=input_dim, max_freq_log2=multires - 1, N_freqs=multires, log_sampling=True)
    elif encoding == "frequency":
        from .freqencoder import FreqEncoder
        encoder = FreqEncoder(input_dim=input_dim, degree=multires)
    elif encoding == "sh":
        from .shencoder import SHEncoder
        encoder = SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding in ("hashgrid", "tiledgrid"):
        from .gridencoder import GridEncoder
        encoder = GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim,
                              base_resolution=base_resolution, log2_hashmap_size=log2_hashmap_size,
                              desired_resolution=desired_resolution,
                              gridtype="hash" if encoding == "hashgrid" else "tiled",
                              align_corners=align_corners, interpolation=interpolation)
    else:
        raise NotImplementedError("Unknown encoding mode, choose from [None, frequency, sh, hashgrid, tiledgrid]")
    return encoder, encoder.output_dim
