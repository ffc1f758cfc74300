"""Audio-to-deformation motion networks of InsTaG, re-implemented for the MI355X path.

Counterpart of /root/reference/scene/motion_net.py (same module / parameter names so reference
``state_dict``s load unchanged):
  AudioAttNet :29-64, AudioNet :67-99, MLP :152-173,
  MotionNetwork (UMF) :176-345, MouthMotionNetwork :346-478, PersonalizedMotionNetwork (PMF) :562-748.
The tri-plane grid encoders are ``instag_amd.gridencoder.GridEncoder`` (HIP); ``encoder_cls`` lets
the CPU tests inject the oracle encoder.  The per-Gaussian MLP chains (B = N rows) run as fused
f32-MFMA HIP kernels through ``instag_amd.mlp``.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

CONCURRENT_AUDIO = True       # fork the audio branch onto a second stream (parallel hipGraph branch)


def _side_stream(device, index=0):
    from . import _lib
    return _lib.side_stream(device, ("frame_branch", index))


class LazyOutputs(dict):
    """dict whose values may be zero-argument callables, evaluated (once) on first access.  The fused render
    path consumes only the raw head outputs ``_h`` / ``_p``; the derived entries of the reference's result dict
    (d_xyz = h[:, :3] * 1e-2, p_scale = tanh(.)*0.25+1, ...) cost a launch each and are built on demand."""

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if callable(v) and not torch.is_tensor(v):
            v = v()
            dict.__setitem__(self, k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


_AUDIO_DIMS = (("esperanto", 44), ("deepspeech", 29), ("hubert", 1024), ("ave", 32))


def audio_in_dim(extractor: str) -> int:
    for key, dim in _AUDIO_DIMS:
        if key in extractor:
            return dim
    raise NotImplementedError(extractor)


class AudioAttNet(nn.Module):
    """Attention over the 8-frame audio window -> one feature vector (motion_net.py:29-64)."""

    def __init__(self, dim_aud=64, seq_len=8):
        super().__init__()
        self.seq_len, self.dim_aud = seq_len, dim_aud
        chans = [dim_aud, 16, 8, 4, 2, 1]
        layers = []
        for cin, cout in zip(chans[:-1], chans[1:]):
            layers += [nn.Conv1d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True), nn.LeakyReLU(0.02, True)]
        self.attentionConvNet = nn.Sequential(*layers)
        self.attentionNet = nn.Sequential(nn.Linear(seq_len, seq_len, bias=True), nn.Softmax(dim=1))

    def forward(self, x):                       # x: [1, seq_len, dim_aud]
        y = self.attentionConvNet(x.permute(0, 2, 1))
        y = self.attentionNet(y.view(1, self.seq_len)).view(1, self.seq_len, 1)
        return torch.sum(y * x, dim=1)


class AudioNet(nn.Module):
    """Per-frame audio feature encoder: 4 stride-2 conv1d + 2 FC (motion_net.py:67-99)."""

    def __init__(self, dim_in=29, dim_aud=64, win_size=16):
        super().__init__()
        self.win_size, self.dim_aud = win_size, dim_aud
        mid = 32 if dim_in < 128 else 128
        chans = [dim_in, mid, mid, 64, 64]
        layers = []
        for cin, cout in zip(chans[:-1], chans[1:]):
            layers += [nn.Conv1d(cin, cout, kernel_size=3, stride=2, padding=1, bias=True), nn.LeakyReLU(0.02, True)]
        self.encoder_conv = nn.Sequential(*layers)
        self.encoder_fc1 = nn.Sequential(nn.Linear(64, 64), nn.LeakyReLU(0.02, True), nn.Linear(64, dim_aud))

    def forward(self, x):
        half = self.win_size // 2
        x = x[:, :, 8 - half:8 + half]
        return self.encoder_fc1(self.encoder_conv(x).squeeze(-1))


class MLP(nn.Module):
    """Bias-free ReLU MLP (motion_net.py:152-173)."""

    def __init__(self, dim_in, dim_out, dim_hidden, num_layers):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hidden, self.num_layers = dim_in, dim_out, dim_hidden, num_layers
        self.net = nn.ModuleList([
            nn.Linear(dim_in if l == 0 else dim_hidden, dim_out if l == num_layers - 1 else dim_hidden, bias=False)
            for l in range(num_layers)])

    def forward(self, x):
        if x.is_cuda and x.dim() == 2:
            # per-Gaussian matrix on the device: one fused f32-MFMA HIP kernel per pass (instag_amd/mlp.py)
            from . import mlp as _mlp
            if _mlp.supported(self.dim_in, self.dim_hidden, self.dim_out, self.num_layers):
                return _mlp.fused_mlp(x, [layer.weight for layer in self.net])
        # per-frame vectors (e.g. the 5-element expression code) and host-side tests: plain torch ops
        for l, layer in enumerate(self.net):
            x = layer(x)
            if l != self.num_layers - 1:
                x = F.relu(x)
        return x


def _default_encoder_cls():
    from .gridencoder import GridEncoder
    return GridEncoder


class _TriPlaneField(nn.Module):
    """Shared part of UMF / PMF: audio branch, tri-plane encoders, attention MLPs, sigma_net."""

    def __init__(self, audio_extractor, audio_dim, hidden_dim, exp_eye, out_dim, ind_dim=0, encoder_cls=None):
        super().__init__()
        encoder_cls = encoder_cls or _default_encoder_cls()
        self.audio_in_dim = audio_in_dim(audio_extractor)
        self.bound = 0.15
        self.exp_eye = exp_eye
        self.individual_dim = ind_dim
        if ind_dim > 0:
            self.individual_codes = nn.Parameter(torch.randn(10000, ind_dim) * 0.1)
        self.audio_dim = audio_dim
        if audio_extractor == "ave":
            raise NotImplementedError("the 'ave' audio extractor is outside the accelerated path")
        self.audio_net = AudioNet(self.audio_in_dim, audio_dim)
        self.audio_att_net = AudioAttNet(audio_dim)
        self.num_levels, self.level_dim = 12, 1
        enc = dict(input_dim=2, num_levels=self.num_levels, level_dim=self.level_dim, base_resolution=16,
                   log2_hashmap_size=17, desired_resolution=256 * self.bound, gridtype="hash", align_corners=False)
        self.encoder_xy, self.encoder_yz, self.encoder_xz = encoder_cls(**enc), encoder_cls(**enc), encoder_cls(**enc)
        self.in_dim_xy = self.in_dim_yz = self.in_dim_xz = self.encoder_xy.output_dim
        self.in_dim = 3 * self.encoder_xy.output_dim
        self.num_layers = 3
        self.hidden_dim = hidden_dim
        self.exp_in_dim = 5
        self.eye_dim = 6 if exp_eye else 0
        if exp_eye:
            self.exp_encode_net = MLP(self.exp_in_dim, self.eye_dim - 1, 16, 2)
            self.eye_att_net = MLP(self.in_dim, self.eye_dim, 16, 2)
        self.out_dim = out_dim
        self.sigma_net = MLP(self.in_dim + audio_dim + self.eye_dim + ind_dim, out_dim, hidden_dim, self.num_layers)
        self.aud_ch_att_net = MLP(self.in_dim, audio_dim, 32, 2)

    def encode_x(self, xyz, bound, shift=None):
        """``shift`` = (tensor [N, >=3], scale): encode xyz + scale * tensor[:, :3] (fused into the kernel on the GPU)."""
        if xyz.is_cuda:
            from . import gridencoder as _ge
            if _ge.tri_plane_supported(self.encoder_xy, self.encoder_yz, self.encoder_xz):
                # the three planes in one HIP kernel, output already concatenated
                if shift is not None:
                    return _ge.tri_plane_encode(xyz, self.encoder_xy, self.encoder_yz, self.encoder_xz, bound,
                                                shift[0], shift[1])
                return _ge.tri_plane_encode(xyz, self.encoder_xy, self.encoder_yz, self.encoder_xz, bound)
        if shift is not None:
            xyz = torch.add(xyz, shift[0][..., :3], alpha=shift[1])
        xy, yz = xyz[:, :-1], xyz[:, 1:]
        xz = torch.cat([xyz[:, :1], xyz[:, -1:]], dim=-1)
        return torch.cat([self.encoder_xy(xy, bound=bound), self.encoder_yz(yz, bound=bound),
                          self.encoder_xz(xz, bound=bound)], dim=-1)

    def encode_audio(self, a):
        if a is None:
            return None
        return self.audio_att_net(self.audio_net(a).unsqueeze(0))

    def encode_exp(self, e):
        return torch.cat([self.exp_encode_net(e[:-1]), e[-1:]], dim=-1)

    def encode_frame(self, a, e):
        """-> (enc_a [1, audio_dim], enc_e [6] or None): everything that depends on the frame only."""
        want_e = self.exp_eye and e is not None
        if a is not None and a.is_cuda:
            from . import audio as _audio
            if _audio.supported(self, a, e if want_e else None):
                # AudioNet + AudioAttNet + expression MLP as one workgroup per pass (instag_amd/audio.py)
                return _audio.frame_codes(self, a, e if want_e else None)
        return self.encode_audio(a), (self.encode_exp(e) if want_e else None)

    def start_audio(self, a, stream_index=0, e=None):
        """Mark the point of the step from which the per-frame branch may run: the next forward(x, a, e) with the
        same `a` launches it on side stream `stream_index` behind an event recorded HERE, so on the device it
        overlaps everything the caller enqueues in between, while in the autograd graph it is created late
        (right before the glue that consumes it) and its backward is therefore scheduled early."""
        from . import _lib
        if not (a.is_cuda and CONCURRENT_AUDIO and _lib.may_fork(a.device)):
            return
        ev = self.__dict__.get("_audio_event")      # one reusable event per network (no create / destroy per step:
        if ev is None:                              # destroying an event while a stream capture is open aborts)
            ev = self.__dict__["_audio_event"] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(a.device))
        self._audio_pending = (a, e, ev, _side_stream(a.device, stream_index))

    def _trunk(self, x, a, e, c, enc_x=None, x_shift=None):
        """-> (enc_x, ambient_aud [N,1], ambient_eye [N,1] or None, h [N,out_dim], amb3); amb3 = the [N,3] tensor
        (aud, eye, 0) the two ambient columns are views of (fused path) or None.  (Nothing of a step may be kept
        on the module: a live autograd graph across steps breaks stream capture.)"""
        from . import _lib
        fork = x.is_cuda and CONCURRENT_AUDIO and _lib.may_fork(x.device)
        pending = self.__dict__.pop("_audio_pending", None)
        if enc_x is None:
            enc_x = self.encode_x(x, bound=self.bound, shift=x_shift)
        aud_ch_att = eye_pre = None
        if self.exp_eye and enc_x.is_cuda and enc_x.dim() == 2:
            from . import mlp as _mlp
            na, ne = self.aud_ch_att_net, self.eye_att_net
            if na.num_layers == 2 and ne.num_layers == 2 and \
                    _mlp.supported(na.dim_in, na.dim_hidden, na.dim_out, 2) and \
                    _mlp.supported(ne.dim_in, ne.dim_hidden, ne.dim_out, 2):
                # both attention MLPs read enc_x, and so does sigma_net's input below: one operator whose backward
                # kernels sum the three gradients of enc_x (instag_amd/mlp.py:_SharedInputMLPs)
                aud_ch_att, eye_pre, enc_x = _mlp.shared_input_mlps(
                    enc_x, [l.weight for l in na.net], [l.weight for l in ne.net])
        if aud_ch_att is None:
            aud_ch_att = self.aud_ch_att_net(enc_x)
            eye_pre = self.eye_att_net(enc_x) if self.exp_eye else None
        if fork:
            # the per-frame branch only meets the per-Gaussian branch at the glue: it runs on a second stream,
            # gated by the early event when the caller announced the frame with start_audio()
            main_stream = torch.cuda.current_stream(x.device)
            if pending is not None and pending[0] is a and pending[1] is e:
                side = pending[3]
                side.wait_event(pending[2])
            else:
                side = _side_stream(x.device)
                side.wait_stream(main_stream)
            with torch.cuda.stream(side):
                enc_a, enc_e = self.encode_frame(a, e)
            main_stream.wait_stream(side)
            from . import _keepalive
            _keepalive.cross_stream(enc_a, main_stream)
            _keepalive.cross_stream(enc_e, main_stream)
        else:
            enc_a, enc_e = self.encode_frame(a, e)
        if self.exp_eye and c is None and enc_x.is_cuda:
            from . import glue as _glue
            if _glue.motion_glue_supported(enc_x, aud_ch_att, eye_pre):
                # repeat / mul / relu / cat / norm chain as one HIP kernel per pass (instag_amd/glue.py)
                # (also without gradients -- inference, the mouth branch's jaw feature: the same two launches, and
                # the forward-only form of the kernels keeps no activations)
                if _glue.glue_sigma_supported(enc_x, aud_ch_att, eye_pre, self.sigma_net):
                    # glue + sigma_net as one autograd node: its backward is ONE kernel (instag_amd/glue.py:_GlueSigma)
                    h, amb = _glue.glue_sigma(enc_x, aud_ch_att, eye_pre, enc_a, enc_e, self.sigma_net,
                                              frame_stream=side if fork else None)
                    return enc_x, amb[:, 0:1], amb[:, 1:2], h, amb
                h_in, amb = _glue.motion_glue(enc_x, aud_ch_att, eye_pre, enc_a, enc_e,
                                              frame_stream=side if fork else None)
                return enc_x, amb[:, 0:1], amb[:, 1:2], self.sigma_net(h_in), amb
        parts = [enc_x, enc_a.repeat(enc_x.shape[0], 1) * aud_ch_att]
        eye_att = None
        if self.exp_eye:
            eye_att = torch.relu(eye_pre)
            parts.append(enc_e * eye_att)
        if c is not None:
            parts.append(c.repeat(enc_x.shape[0], 1))
        h = self.sigma_net(torch.cat(parts, dim=-1))
        amb_eye = eye_att.norm(dim=-1, keepdim=True) if eye_att is not None else None
        return enc_x, aud_ch_att.norm(dim=-1, keepdim=True), amb_eye, h, None


class MotionNetwork(_TriPlaneField):
    """Universal motion field (UMF), hidden 64, 11 outputs (motion_net.py:176-345)."""

    def __init__(self, audio_dim=32, ind_dim=0, args=None, encoder_cls=None):
        super().__init__(args.audio_extractor, audio_dim, 64, True, 11, ind_dim, encoder_cls)
        self.cache = None

    def forward(self, x, a, e=None, c=None, x_shift=None):
        """Reference signature (x, a, e, c).  Extension: ``x_shift`` = (tensor, scale) evaluates the field at
        x + scale * tensor[:, :3] without materialising the sum (render_motion's xyz + p_xyz)."""
        _, amb_aud, amb_eye, h, amb3 = self._trunk(x, a, e, c, x_shift=x_shift)
        def outputs(h, amb_aud, amb_eye):
            return LazyOutputs({
                "d_xyz": lambda: h[..., :3] * 1e-2, "d_rot": h[..., 3:7], "d_opa": h[..., 7:8],
                "d_scale": h[..., 8:11], "ambient_aud": amb_aud, "ambient_eye": amb_eye,
                "_h": h,          # raw head output, consumed by the fused deform / regulariser operators
            })

        results = outputs(h, amb_aud, amb_eye)
        results["_amb3"] = amb3            # (ambient_aud, ambient_eye, 0) as one tensor = the attention colours
        # consumed without gradients by the mouth branch at inference (gaussian_renderer/__init__.py:362-363);
        # detached so that a finished step does not keep its autograd graph (and grad accumulators) alive
        self.cache = outputs(h.detach(), amb_aud.detach(), None if amb_eye is None else amb_eye.detach())
        return results

    def get_params(self, lr, lr_net, wd=0):
        params = [
            {"params": self.audio_net.parameters(), "lr": lr_net, "weight_decay": wd},
            {"params": self.encoder_xy.parameters(), "lr": lr},
            {"params": self.encoder_yz.parameters(), "lr": lr},
            {"params": self.encoder_xz.parameters(), "lr": lr},
            {"params": self.sigma_net.parameters(), "lr": lr_net, "weight_decay": wd},
            {"params": self.audio_att_net.parameters(), "lr": lr_net * 5, "weight_decay": 0.0001},
        ]
        if self.individual_dim > 0:
            params.append({"params": self.individual_codes, "lr": lr_net, "weight_decay": wd})
        params += [
            {"params": self.aud_ch_att_net.parameters(), "lr": lr_net, "weight_decay": wd},
            {"params": self.eye_att_net.parameters(), "lr": lr_net, "weight_decay": wd},
            {"params": self.exp_encode_net.parameters(), "lr": lr_net, "weight_decay": wd},
        ]
        return params


class PersonalizedMotionNetwork(_TriPlaneField):
    """Personalised motion field (PMF) + alignment head (motion_net.py:562-748)."""

    def __init__(self, audio_dim=32, ind_dim=0, args=None, encoder_cls=None):
        face = args.type == "face"
        super().__init__(args.audio_extractor, audio_dim, 32 if face else 16, face, 11 if face else 7, ind_dim,
                         encoder_cls)
        self.args = args
        self.align_net = MLP(self.in_dim, 6, self.hidden_dim, 2)

    def forward(self, x, a, e=None, c=None, va=None):
        """Same result dict as the reference.  Only the alignment head (p_xyz, p_scale) is evaluated eagerly: the
        deformation head (audio branch, attention MLPs, sigma_net) runs on first access of one of its entries,
        so a caller that renders with ``personalized=False`` (train_face.py warm stage) does not pay for outputs
        it never reads -- the values, when read, are the same."""
        face = self.args.type == "face"
        enc_x = self.encode_x(x, bound=self.bound)
        p = self.align_net(enc_x)
        memo = {}

        def head():
            if "r" not in memo:
                memo["r"] = self._trunk(x, a, e, c, enc_x=enc_x)
            return memo["r"]            # (enc_x, amb_aud, amb_eye, h, amb3)

        return LazyOutputs({
            "d_xyz": lambda: head()[3][..., :3] * 1e-2, "d_rot": lambda: head()[3][..., 3:7],
            "d_opa": (lambda: head()[3][..., 7:8]) if face else None,
            "d_scale": (lambda: head()[3][..., 8:11]) if face else None,
            "ambient_aud": lambda: head()[1],
            "ambient_eye": (lambda: head()[2]) if self.exp_eye else None,
            "p_xyz": lambda: p[..., :3] * 1e-2,
            "p_scale": lambda: torch.tanh(p[..., 3:] / 5) * 0.25 + 1,
            "_h": lambda: head()[3], "_p": p, "_amb3": lambda: head()[4],
        })

    def get_params(self, lr, lr_net, wd=0):
        params = [
            {"params": self.audio_net.parameters(), "name": "neural_audio_net", "lr": lr_net, "weight_decay": wd},
            {"params": self.encoder_xy.parameters(), "name": "neural_encoder_xy", "lr": lr},
            {"params": self.encoder_yz.parameters(), "name": "neural_encoder_xy", "lr": lr},
            {"params": self.encoder_xz.parameters(), "name": "neural_encoder_xy", "lr": lr},
            {"params": self.sigma_net.parameters(), "name": "neural_sigma_net", "lr": lr_net, "weight_decay": wd},
            {"params": self.align_net.parameters(), "name": "neural_align_net", "lr": lr_net / 2, "weight_decay": wd},
            {"params": self.audio_att_net.parameters(), "name": "neural_audio_att_net", "lr": lr_net * 5,
             "weight_decay": 0.0001},
        ]
        if self.individual_dim > 0:
            params.append({"params": self.individual_codes, "name": "neural_individual_codes", "lr": lr_net,
                           "weight_decay": wd})
        params.append({"params": self.aud_ch_att_net.parameters(), "name": "neural_aud_ch_att_net", "lr": lr_net,
                       "weight_decay": wd})
        if self.exp_eye:
            params.append({"params": self.eye_att_net.parameters(), "name": "neural_eye_att_net", "lr": lr_net,
                           "weight_decay": wd})
            params.append({"params": self.exp_encode_net.parameters(), "name": "neural_exp_encode_net",
                           "lr": lr_net, "weight_decay": wd})
        return params


class MouthMotionNetwork(nn.Module):
    """Motion field of the mouth branch (scene/motion_net.py:346-478): tri-plane encoders with base resolution 64
    (46,600 entries per plane: too large for LDS, so the tri-plane kernels read the tables in place --
    csrc/grid.hip triplane_global_*), sigma_net 71->32->32->7 on the concatenation of position code, audio code and the
    3-element jaw-movement feature, scaler_net 39->16->16->1 gating the displacement."""

    XYZ_SCALE = (1e-2 / 5, 1e-2, 1e-2 / 5)      # per-axis displacement scale (scene/motion_net.py:446-450)

    def __init__(self, audio_dim=32, ind_dim=0, args=None, encoder_cls=None):
        super().__init__()
        encoder_cls = encoder_cls or _default_encoder_cls()
        self.audio_in_dim = audio_in_dim(args.audio_extractor)
        self.bound = 0.15
        self.individual_dim = ind_dim
        if ind_dim > 0:
            self.individual_codes = nn.Parameter(torch.randn(10000, ind_dim) * 0.1)
        self.audio_dim = audio_dim
        if args.audio_extractor == "ave":
            raise NotImplementedError("the 'ave' audio extractor is outside the accelerated path")
        self.audio_net = AudioNet(self.audio_in_dim, audio_dim)
        self.audio_att_net = AudioAttNet(audio_dim)
        self.num_levels, self.level_dim = 12, 1
        enc = dict(input_dim=2, num_levels=self.num_levels, level_dim=self.level_dim, base_resolution=64,
                   log2_hashmap_size=17, desired_resolution=384 * self.bound, gridtype="hash", align_corners=False)
        self.encoder_xy, self.encoder_yz, self.encoder_xz = encoder_cls(**enc), encoder_cls(**enc), encoder_cls(**enc)
        self.in_dim_xy = self.in_dim_yz = self.in_dim_xz = self.encoder_xy.output_dim
        self.in_dim = 3 * self.encoder_xy.output_dim
        self.num_layers, self.hidden_dim, self.out_dim, self.move_dim = 3, 32, 7, 3
        self.sigma_net = MLP(self.in_dim + audio_dim + ind_dim + self.move_dim, self.out_dim, self.hidden_dim,
                             self.num_layers)
        self.scaler_net = MLP(self.in_dim + self.move_dim, 1, 16, 3)
        self.aud_ch_att_net = MLP(self.in_dim, audio_dim, 32, 2)     # present in the reference, unused in forward
        # d_xyz = h[:, :3] * 1e-2 with x and z divided by 5 (motion_net.py:446-450); not part of the state_dict
        self.register_buffer("_xyz_scale", torch.tensor(self.XYZ_SCALE), persistent=False)

    encode_x = _TriPlaneField.encode_x

    def encode_audio(self, a):
        if a is None:
            return None
        if a.is_cuda:
            from . import audio as _audio
            if _audio.supported(self, a, None):
                return _audio.frame_codes(self, a, None)[0]       # AudioNet + AudioAttNet in one workgroup
        return self.audio_att_net(self.audio_net(a).unsqueeze(0))

    def start_audio(self, a, stream_index=0):
        """As _TriPlaneField.start_audio: the next forward(x, a, move) with the same `a` runs the audio branch on side
        stream `stream_index` behind an event recorded here."""
        from . import _lib
        if not (a.is_cuda and CONCURRENT_AUDIO and _lib.may_fork(a.device)):
            return
        ev = self.__dict__.get("_audio_event")
        if ev is None:
            ev = self.__dict__["_audio_event"] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(a.device))
        self._audio_pending = (a, ev, _side_stream(a.device, stream_index))

    def forward(self, x, a, move, x_shift=None):
        """Reference signature (x, a, move).  Extension: ``x_shift`` = (tensor, scale) evaluates the field at
        x + scale * tensor[:, :3] without materialising the sum (the mouth render's xyz + p_xyz)."""
        pending = self.__dict__.pop("_audio_pending", None)
        enc_x = self.encode_x(x, bound=self.bound, shift=x_shift)
        if pending is not None and pending[0] is a:
            # the audio branch depends on the frame only: on its own stream from the point the caller announced it
            main_stream, side = torch.cuda.current_stream(a.device), pending[2]
            side.wait_event(pending[1])
            with torch.cuda.stream(side):
                enc_a = self.encode_audio(a)
            main_stream.wait_stream(side)
            from . import _keepalive
            _keepalive.cross_stream(enc_a, main_stream)
        else:
            enc_a = self.encode_audio(a)
        n = enc_x.shape[0]
        from . import glue as _glue
        if enc_x.is_cuda and _glue.mouth_glue_supported(enc_x, enc_a, move):
            in_sigma, in_scaler = _glue.mouth_glue(enc_x, enc_a, move)        # both inputs in one launch per pass
        else:
            move = move.repeat(n, 1)
            in_sigma = torch.cat([enc_x, enc_a.repeat(n, 1), move], dim=-1)
            in_scaler = torch.cat([enc_x, move], dim=-1)
        h = self.sigma_net(in_sigma)
        h_s = self.scaler_net(in_scaler)
        # d_xyz = h[:, :3] * scale (the reference's in-place edits as one multiply) gated by sigmoid(h_s) * 2; the fused
        # render path consumes the raw head outputs (glue.mouth_activate), the dictionary entries are built on access
        return LazyOutputs({"d_xyz": lambda: (h[..., :3] * self._xyz_scale) * torch.sigmoid(h_s) * 2,
                            "d_rot": lambda: h[..., 3:], "_h": h, "_hs": h_s})

    def get_params(self, lr, lr_net, wd=0):
        params = [
            {"params": self.audio_net.parameters(), "lr": lr_net, "weight_decay": wd},
            {"params": self.encoder_xy.parameters(), "lr": lr},
            {"params": self.encoder_yz.parameters(), "lr": lr},
            {"params": self.encoder_xz.parameters(), "lr": lr},
            {"params": self.sigma_net.parameters(), "lr": lr_net, "weight_decay": wd},
            {"params": self.scaler_net.parameters(), "lr": lr_net, "weight_decay": wd},
            {"params": self.audio_att_net.parameters(), "lr": lr_net * 5, "weight_decay": 0.0001},
        ]
        if self.individual_dim > 0:
            params.append({"params": self.individual_codes, "lr": lr_net, "weight_decay": wd})
        params.append({"params": self.aud_ch_att_net.parameters(), "lr": lr_net, "weight_decay": wd})
        return params

