"""``GridRenderer`` of the reference (scene/neural_renderer.py:49-222): a 3-D hash-grid encoder (16 levels x 2, table
2^19) + SH direction encoder + two small MLPs mapping (position, direction) to (density, colour).

InsTaG constructs it with every Gaussian model (scene/gaussian_model.py:317), registers its parameters in Adam (:394)
and stores its ``state_dict`` in every checkpoint (:130), but never calls ``forward`` -- so what
matters for a drop-in is that it constructs, exposes ``get_params``, and that checkpoints round-trip: same attribute
names, same ``state_dict`` keys and shapes (``bound``, ``coord_center``, ``encoder_x.embeddings``,
``encoder_x.offsets``, ``sigma_net.net.{0,1,2}.weight``, ``color_net.net.{0,1}.weight``).  ``forward`` works all the same
(grid / SH encoders and MLPs are the HIP operators).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .gridencoder import GridEncoder
from .motion_net import MLP
from .shencoder import SHEncoder


class GridRenderer(nn.Module):
    def __init__(self, bound=1., coord_center=(0., 0., 0.), keep_sigma=False):
        super().__init__()
        self.register_buffer("bound", torch.as_tensor(bound, dtype=torch.float32).detach().clone())
        self.register_buffer("coord_center", torch.as_tensor(coord_center, dtype=torch.float32).detach().clone())
        self.keep_sigma = keep_sigma
        self.sigma_results_static = None
        self.num_levels, self.level_dim, self.base_resolution = 16, 2, 16
        self.table_size, self.desired_resolution = 19, 512
        self.encoder_x, self.in_dim_x = self.create_encoder()
        self.num_layers, self.hidden_dim, self.geo_feat_dim = 3, 64, 64
        self.sigma_net = MLP(self.in_dim_x, 1 + self.geo_feat_dim, self.hidden_dim, self.num_layers)
        self.num_layers_color, self.hidden_dim_color = 2, 64
        self.encoder_dir = SHEncoder(input_dim=3, degree=4)
        self.in_dim_dir = self.encoder_dir.output_dim
        self.color_net = MLP(self.in_dim_dir + self.geo_feat_dim, 3, self.hidden_dim_color, self.num_layers_color)

    def create_encoder(self):
        """neural_renderer.py:120-131: the table shapes depend on ``bound`` (desired_resolution * bound)."""
        enc = GridEncoder(input_dim=3, num_levels=self.num_levels, level_dim=self.level_dim,
                          base_resolution=self.base_resolution, log2_hashmap_size=self.table_size,
                          desired_resolution=float(self.desired_resolution * self.bound.cpu()))
        self.encoder_x, self.in_dim_x = enc, enc.output_dim
        return self.encoder_x, self.in_dim_x

    def recover_from_ckpt(self, state_dict):
        """neural_renderer.py:133-139: rebuild the encoder for the checkpoint's bound, then load everything."""
        self.bound = torch.as_tensor(state_dict["bound"], dtype=torch.float32).detach().clone().to(self.bound.device)
        self.encoder_x, self.in_dim_x = self.create_encoder()
        self.load_state_dict(state_dict)

    def encode_x(self, x):
        return self.encoder_x(x - self.coord_center, bound=float(self.bound))

    def density(self, x, enc_x=None):
        if self.keep_sigma and self.sigma_results_static is not None:
            return self.sigma_results_static
        if enc_x is None:
            enc_x = self.encode_x(x)
        h = self.sigma_net(enc_x)
        out = {"sigma": h[..., 0], "geo_feat": h[..., 1:]}
        if self.keep_sigma:
            self.sigma_results_static = out
        return out

    def color(self, sigma_result, d):
        h = torch.cat([self.encoder_dir(d), sigma_result["geo_feat"]], dim=-1)
        return torch.sigmoid(self.color_net(h)) * (1 + 2 * 0.001) - 0.001

    def forward(self, x, d):
        sigma_result = self.density(x, self.encode_x(x))
        return sigma_result["sigma"], self.color(sigma_result, d)

    def get_params(self, lr, lr_net, wd=0):
        return [
            {"params": self.encoder_x.parameters(), "name": "neural_encoder", "lr": lr},
            {"params": self.sigma_net.parameters(), "name": "neural_sigma", "lr": lr_net, "weight_decay": wd},
            {"params": self.color_net.parameters(), "name": "neural_color", "lr": lr_net, "weight_decay": wd},
        ]
