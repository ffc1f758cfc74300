"""CPU oracle: multiresolution grid ("hashgrid") encoder, forward + backward (numpy, fp32).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.

Restates /root/reference/gridencoder/src/gridencoder.cu:
  fast_hash / get_grid_index   :50-84    dense index while stride <= hashmap_size, else prime-xor hash; % size
  kernel_grid                  :87-245   scale = exp2f(l*S)*H - 1, pos = x*scale + 0.5 (align_corners: +0),
                                          floor / fract, optional smoothstep, 2^D corner multilinear sum,
                                          out-of-[0,1] input -> zeros, dy_dx[b, l, d, c]
  kernel_grid_backward         :248-340  grad_grid[index] += w * grad  (atomicAdd)
  kernel_input_backward        :343-369  grad_inputs[b,d] = sum_{l,c} grad[l,b,c] * dy_dx[b,l,d,c]
  kernel_grad_tv               :506-610  per sample and level: vertex v = floor(x*scale + 0.5); over its 2D axis
                                          neighbours n (inside [0, resolution]): r = sum (e[v]-e[n]), q = sum (e[v]-e[n])^2;
                                          grad[v] += weight/(2D) * r * rsqrt(q + 1e-9)
and the host-side parameterisation of /root/reference/gridencoder/grid.py:
  per_level_scale / offsets    :101-128
  input mapping (x+bound)/(2 bound) :149, outputs [L,B,C] -> [B, L*C] :57
Known answers used to pin it (tests/test_oracle_grid.py): the offsets tables of
SURVEY.md §0.3, value at a vertex = its embedding, bilinear midpoint = mean of
4 corners, out-of-range input -> 0.
"""
from __future__ import annotations

import numpy as np

PRIMES = np.array([1, 2654435761, 805459861, 3674653429, 2097192037, 1434869437, 2165219737], dtype=np.uint32)


def per_level_scale(base_resolution, desired_resolution, num_levels):
    return np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))


def make_offsets(input_dim, num_levels, base_resolution, log2_hashmap_size, pls, align_corners=False):
    """gridencoder/grid.py:118-128."""
    offsets, offset = [], 0
    max_params = 2 ** log2_hashmap_size
    for i in range(num_levels):
        resolution = int(np.ceil(base_resolution * pls ** i))
        params = min(max_params, (resolution if align_corners else resolution + 1) ** input_dim)
        params = int(np.ceil(params / 8) * 8)
        offsets.append(offset)
        offset += params
    offsets.append(offset)
    return np.array(offsets, dtype=np.int32)


def _grid_index(gridtype, align_corners, hashmap_size, resolution, pos_grid):
    """pos_grid: [B, D] uint32 -> [B] uint32 index (before *C)."""
    B, D = pos_grid.shape
    stride = np.uint64(1)
    index = np.zeros(B, dtype=np.uint32)
    d = 0
    while d < D and stride <= hashmap_size:
        index = (index + pos_grid[:, d] * np.uint32(stride & np.uint64(0xFFFFFFFF))).astype(np.uint32)
        stride = stride * np.uint64(resolution if align_corners else resolution + 1)
        stride &= np.uint64(0xFFFFFFFF)          # uint32 wrap like the kernel
        d += 1
    if gridtype == 0 and stride > hashmap_size:
        h = np.zeros(B, dtype=np.uint32)
        for i in range(D):
            h ^= (pos_grid[:, i] * PRIMES[i]).astype(np.uint32)
        index = h
    return index % np.uint32(hashmap_size)


def _level_setup(inputs, level, S, H, align_corners, interp):
    scale = np.float32(np.exp2(np.float32(level * np.float32(S))) * np.float32(H) - np.float32(1.0))
    resolution = int(np.ceil(scale)) + 1
    pos = inputs * scale + np.float32(0.0 if align_corners else 0.5)
    pos_grid_f = np.floor(pos)
    pos_grid = pos_grid_f.astype(np.int64).astype(np.uint32)
    frac = (pos - pos_grid_f).astype(np.float32)
    if interp == 1:
        deriv = (6 * frac * (1.0 - frac)).astype(np.float32)
        frac = (frac * frac * (3.0 - 2.0 * frac)).astype(np.float32)
    else:
        deriv = np.ones_like(frac)
    return scale, resolution, pos_grid, frac, deriv


def grid_encode_forward(inputs, embeddings, offsets, S, H, calc_grad_inputs=False,
                        gridtype=0, align_corners=False, interp=0):
    """inputs [B,D] in [0,1]; embeddings [sO,C]; -> outputs [L,B,C], dy_dx [B, L*D*C] or None."""
    inputs = np.ascontiguousarray(inputs, dtype=np.float32)
    B, D = inputs.shape
    C = embeddings.shape[1]
    L = len(offsets) - 1
    with np.errstate(over="ignore"):
        return _forward(inputs, embeddings, offsets, S, H, calc_grad_inputs, gridtype, align_corners, interp, B, D, C, L)


def _forward(inputs, embeddings, offsets, S, H, calc, gridtype, align_corners, interp, B, D, C, L):
    outputs = np.zeros((L, B, C), dtype=np.float32)
    dy_dx = np.zeros((B, L, D, C), dtype=np.float32) if calc else None
    oob = ((inputs < 0) | (inputs > 1)).any(axis=1)
    ok = ~oob
    x = inputs[ok]
    for level in range(L):
        grid = embeddings[offsets[level]:offsets[level + 1]]
        hashmap_size = int(offsets[level + 1] - offsets[level])
        scale, resolution, pos_grid, frac, deriv = _level_setup(x, level, S, H, align_corners, interp)
        res = np.zeros((x.shape[0], C), dtype=np.float32)
        for idx in range(1 << D):
            w = np.ones(x.shape[0], dtype=np.float32)
            pg = pos_grid.copy()
            for d in range(D):
                if idx & (1 << d):
                    w = w * frac[:, d]
                    pg[:, d] = pg[:, d] + np.uint32(1)
                else:
                    w = w * (np.float32(1) - frac[:, d])
            index = _grid_index(gridtype, align_corners, hashmap_size, resolution, pg)
            res = res + w[:, None] * grid[index]
        outputs[level, ok] = res
        if calc:
            for gd in range(D):
                rg = np.zeros((x.shape[0], C), dtype=np.float32)
                for idx in range(1 << (D - 1)):
                    w = np.full(x.shape[0], scale, dtype=np.float32)
                    pg = pos_grid.copy()
                    for nd in range(D - 1):
                        d = nd + 1 if nd >= gd else nd
                        if idx & (1 << nd):
                            w = w * frac[:, d]
                            pg[:, d] = pg[:, d] + np.uint32(1)
                        else:
                            w = w * (np.float32(1) - frac[:, d])
                    left = _grid_index(gridtype, align_corners, hashmap_size, resolution, pg)
                    pg[:, gd] = pg[:, gd] + np.uint32(1)
                    right = _grid_index(gridtype, align_corners, hashmap_size, resolution, pg)
                    rg = rg + w[:, None] * (grid[right] - grid[left]) * deriv[:, gd:gd + 1]
                tmp = dy_dx[:, level, gd, :]
                tmp[ok] = rg
    return outputs, (dy_dx.reshape(B, L * D * C) if calc else None)


def grid_encode_backward(grad, inputs, embeddings, offsets, S, H, dy_dx=None,
                         gridtype=0, align_corners=False, interp=0):
    """grad [L,B,C] -> grad_embeddings [sO,C] (fp64-accumulated), grad_inputs [B,D] or None."""
    inputs = np.ascontiguousarray(inputs, dtype=np.float32)
    B, D = inputs.shape
    C = embeddings.shape[1]
    L = len(offsets) - 1
    ge = np.zeros(embeddings.shape, dtype=np.float64)
    ok = ~((inputs < 0) | (inputs > 1)).any(axis=1)
    x = inputs[ok]
    with np.errstate(over="ignore"):
        for level in range(L):
            hashmap_size = int(offsets[level + 1] - offsets[level])
            scale, resolution, pos_grid, frac, _ = _level_setup(x, level, S, H, align_corners, interp)
            g = grad[level][ok].astype(np.float64)
            for idx in range(1 << D):
                w = np.ones(x.shape[0], dtype=np.float32)
                pg = pos_grid.copy()
                for d in range(D):
                    if idx & (1 << d):
                        w = w * frac[:, d]
                        pg[:, d] = pg[:, d] + np.uint32(1)
                    else:
                        w = w * (np.float32(1) - frac[:, d])
                index = _grid_index(gridtype, align_corners, hashmap_size, resolution, pg).astype(np.int64)
                np.add.at(ge, index + int(offsets[level]), w[:, None].astype(np.float64) * g)
    gi = None
    if dy_dx is not None:
        dd = dy_dx.reshape(B, L, D, C).astype(np.float64)
        gi = np.einsum("lbc,bldc->bd", grad.astype(np.float64), dd).astype(np.float32)
    return ge.astype(np.float32), gi


def grad_total_variation(inputs, embeddings, grad, offsets, weight, S, H, gridtype=0, align_corners=False):
    """gridencoder.cu:506-610 (grid.py:165-185): adds the total-variation gradient of the table entries hit by
    `inputs` [B,D] in [0,1] into `grad` [sO,C] (float64 accumulation; returned as a new float32 array)."""
    inputs = np.ascontiguousarray(inputs, dtype=np.float32)
    B, D = inputs.shape
    C = embeddings.shape[1]
    L = len(offsets) - 1
    out = grad.astype(np.float64).copy()
    ok = ~((inputs < 0) | (inputs > 1)).any(axis=1)
    x = inputs[ok]
    w = np.float32(np.float32(weight) / np.float32(2 * D))
    with np.errstate(over="ignore"):
        for level in range(L):
            grid = embeddings[offsets[level]:offsets[level + 1]]
            hashmap_size = int(offsets[level + 1] - offsets[level])
            _, resolution, pos_grid, _, _ = _level_setup(x, level, S, H, align_corners, 0)
            index = _grid_index(gridtype, align_corners, hashmap_size, resolution, pos_grid).astype(np.int64)
            r = np.zeros((x.shape[0], C), dtype=np.float32)
            q = np.zeros((x.shape[0], C), dtype=np.float32)
            for d in range(D):
                cur = pos_grid[:, d]
                for step, valid in ((1, cur < resolution), (-1, cur > 0)):
                    pg = pos_grid.copy()
                    pg[:, d] = (cur.astype(np.int64) + step).astype(np.uint32)
                    nb = _grid_index(gridtype, align_corners, hashmap_size, resolution, pg).astype(np.int64)
                    gv = (grid[index] - grid[nb]).astype(np.float32) * valid[:, None]
                    r = r + gv
                    q = q + gv * gv
            term = (w * r * (np.float32(1.0) / np.sqrt(q + np.float32(1e-9)))).astype(np.float64)
            np.add.at(out, index + int(offsets[level]), term)
    return out.astype(np.float32)


class GridEncoderRef:
    """Host-side mirror of gridencoder/grid.py:96-161 on numpy, for tests."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale_=2, base_resolution=16,
                 log2_hashmap_size=19, desired_resolution=None, gridtype="hash", align_corners=False,
                 interpolation="linear", seed=0):
        if desired_resolution is not None:
            per_level_scale_ = per_level_scale(base_resolution, desired_resolution, num_levels)
        self.input_dim, self.num_levels, self.level_dim = input_dim, num_levels, level_dim
        self.per_level_scale, self.base_resolution = per_level_scale_, base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype_id = {"hash": 0, "tiled": 1}[gridtype]
        self.interp_id = {"linear": 0, "smoothstep": 1}[interpolation]
        self.align_corners = align_corners
        self.offsets = make_offsets(input_dim, num_levels, base_resolution, log2_hashmap_size,
                                    per_level_scale_, align_corners)
        rng = np.random.default_rng(seed)
        self.embeddings = rng.uniform(-1e-4, 1e-4, size=(int(self.offsets[-1]), level_dim)).astype(np.float32)

    def forward(self, inputs, bound=1, calc_grad_inputs=False):
        x = (np.asarray(inputs, dtype=np.float32) + np.float32(bound)) / np.float32(2 * bound)
        out, dy_dx = grid_encode_forward(x.reshape(-1, self.input_dim), self.embeddings, self.offsets,
                                         np.log2(self.per_level_scale), self.base_resolution,
                                         calc_grad_inputs, self.gridtype_id, self.align_corners, self.interp_id)
        B = out.shape[1]
        return out.transpose(1, 0, 2).reshape(B, self.output_dim), dy_dx
