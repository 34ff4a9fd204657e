"""TEST INFRASTRUCTURE -- CPU oracle: a plain-PyTorch fp32 restatement of the GLARE hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product (glare_amd/) never does.  Every class keeps the reference's parameter names so that a
state-dict produced by the reference loads unchanged, and cites the file:line it restates
(paths relative to /root/reference/code/models/modules/).

Pinning: tests/test_oracle_vs_reference.py loads the *imported* reference modules (build
container only, via oracle/refimport.py) with seeded weights and checks every stage of this file
against them; small tensors from that import are committed under tests/golden/ so the pin also
holds where /root/reference does not exist.  The deformable convolution: the reference implements
it in CUDA only (ops/dcn/src/*), so on the CPU it is pinned by identities (tests/test_dcn_oracle.py);
since round 6 the reference's own extension, built for gfx950 by oracle/build_ref.py, runs on the GPU
box and pins the restatement BY EXECUTION (tests/test_gpu_dcn_reference.py: <= 5.8e-7).  It follows
deform_conv_cuda_kernel.cu:468-497,571-633 and deform_conv_cuda.cpp:490-569 line by line.

Tolerance contract of the float kernels tested against this oracle is written in each test.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------
# VQGAN building blocks                                    encoder_decoder.py
# --------------------------------------------------------------------------------------------
def swish(x):  # encoder_decoder.py:29-31
    return x * torch.sigmoid(x)


def group_norm(c):  # encoder_decoder.py:34-35
    return nn.GroupNorm(32, c, eps=1e-6, affine=True)


class ResnetBlock(nn.Module):  # encoder_decoder.py:78-137 (temb_channels=0, dropout=0)
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = group_norm(cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm2 = group_norm(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        if cin != cout:
            self.nin_shortcut = nn.Conv2d(cin, cout, 1, 1, 0)
        self.cin, self.cout = cin, cout

    def forward(self, x):
        h = self.conv1(swish(self.norm1(x)))
        h = self.conv2(swish(self.norm2(h)))
        if self.cin != self.cout:
            x = self.nin_shortcut(x)
        return x + h


class AttnBlock(nn.Module):  # encoder_decoder.py:140-192
    def __init__(self, c):
        super().__init__()
        self.norm = group_norm(c)
        self.q = nn.Conv2d(c, c, 1)
        self.k = nn.Conv2d(c, c, 1)
        self.v = nn.Conv2d(c, c, 1)
        self.proj_out = nn.Conv2d(c, c, 1)

    def forward(self, x):
        hn = self.norm(x)
        q, k, v = self.q(hn), self.k(hn), self.v(hn)
        b, c, h, w = q.shape
        q = q.reshape(b, c, h * w).permute(0, 2, 1)
        k = k.reshape(b, c, h * w)
        att = torch.bmm(q, k) * (int(c) ** (-0.5))  # [b, i(query), j(key)]
        att = F.softmax(att, dim=2)
        v = v.reshape(b, c, h * w)
        o = torch.bmm(v, att.permute(0, 2, 1)).reshape(b, c, h, w)
        return x + self.proj_out(o)


class Upsample(nn.Module):  # encoder_decoder.py:38-53
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 1, 1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class Downsample(nn.Module):  # encoder_decoder.py:56-75
    def __init__(self, c):
        super().__init__()
        self.conv = nn.Conv2d(c, c, 3, 2, 0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class _Level(nn.Module):
    pass


class Encoder(nn.Module):  # encoder_decoder.py:342-442
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4), num_res_blocks=2, attn_resolutions=(64,),
                 in_channels=3, resolution=256, z_channels=3, double_z=False):
        super().__init__()
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        res = resolution
        in_mult = (1,) + tuple(ch_mult)
        self.down = nn.ModuleList()
        for lvl in range(self.num_resolutions):
            cin, cout = ch * in_mult[lvl], ch * ch_mult[lvl]
            level = _Level()
            level.block = nn.ModuleList()
            level.attn = nn.ModuleList()
            for _ in range(num_res_blocks):
                level.block.append(ResnetBlock(cin, cout))
                cin = cout
                if res in attn_resolutions:
                    level.attn.append(AttnBlock(cin))
            if lvl != self.num_resolutions - 1:
                level.downsample = Downsample(cin)
                res //= 2
            self.down.append(level)
        self.mid = _Level()
        self.mid.block_1 = ResnetBlock(cin, cin)
        self.mid.attn_1 = AttnBlock(cin)
        self.mid.block_2 = ResnetBlock(cin, cin)
        self.norm_out = group_norm(cin)
        self.conv_out = nn.Conv2d(cin, 2 * z_channels if double_z else z_channels, 3, 1, 1)

    def forward(self, x, mid_feat=False):
        feats = []
        h = self.conv_in(x)
        for lvl in range(self.num_resolutions):
            level = self.down[lvl]
            for i in range(self.num_res_blocks):
                h = level.block[i](h)
                if len(level.attn) > 0:
                    h = level.attn[i](h)
            if lvl != self.num_resolutions - 1:
                feats.append(h)
                h = level.downsample(h)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        h = self.conv_out(swish(self.norm_out(h)))
        return (h, feats) if mid_feat else h


def _decoder_trunk(mod, ch, ch_mult, num_res_blocks, attn_resolutions, resolution, z_channels):
    """Shared constructor of Decoder / MultiScaleDecoder2 (identical trunks:
    encoder_decoder.py:445-513 and deformableDecoder_arch.py:413-482)."""
    n = len(ch_mult)
    mod.num_resolutions, mod.num_res_blocks = n, num_res_blocks
    cin = ch * ch_mult[n - 1]
    res = resolution // 2 ** (n - 1)
    mod.conv_in = nn.Conv2d(z_channels, cin, 3, 1, 1)
    mod.mid = _Level()
    mod.mid.block_1 = ResnetBlock(cin, cin)
    mod.mid.attn_1 = AttnBlock(cin)
    mod.mid.block_2 = ResnetBlock(cin, cin)
    ups = []
    for lvl in reversed(range(n)):
        cout = ch * ch_mult[lvl]
        level = _Level()
        level.block = nn.ModuleList()
        level.attn = nn.ModuleList()
        for _ in range(num_res_blocks + 1):
            level.block.append(ResnetBlock(cin, cout))
            cin = cout
            if res in attn_resolutions:
                level.attn.append(AttnBlock(cin))
        if lvl != 0:
            level.upsample = Upsample(cin)
            res *= 2
        ups.insert(0, level)
    mod.up = nn.ModuleList(ups)
    mod.norm_out = group_norm(cin)
    return cin


class Decoder(nn.Module):  # encoder_decoder.py:445-551
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4), num_res_blocks=2, attn_resolutions=(64,),
                 resolution=256, z_channels=3):
        super().__init__()
        cin = _decoder_trunk(self, ch, ch_mult, num_res_blocks, attn_resolutions, resolution, z_channels)
        self.conv_out = nn.Conv2d(cin, out_ch, 3, 1, 1)

    def forward(self, z):
        feats = []
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for lvl in reversed(range(self.num_resolutions)):
            level = self.up[lvl]
            for i in range(self.num_res_blocks + 1):
                h = level.block[i](h)
                if len(level.attn) > 0:
                    h = level.attn[i](h)
            if lvl != 2:  # encoder_decoder.py:538-539 (hard-coded level index)
                feats.append(h)
            if lvl != 0:
                h = level.upsample(h)
        h = self.conv_out(swish(self.norm_out(h)))
        return h, feats


# --------------------------------------------------------------------------------------------
# Codebook                                                  quantize.py:213-329
# --------------------------------------------------------------------------------------------
class VectorQuantizer2(nn.Module):
    def __init__(self, n_e, e_dim, beta, legacy=False):
        super().__init__()
        self.n_e, self.e_dim, self.beta, self.legacy = n_e, e_dim, beta, legacy
        self.embedding = nn.Embedding(n_e, e_dim)
        self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)

    def forward(self, z):
        zp = z.permute(0, 2, 3, 1).contiguous()  # quantize.py:276
        flat = zp.view(-1, self.e_dim)
        e = self.embedding.weight
        d = torch.sum(flat ** 2, dim=1, keepdim=True) + torch.sum(e ** 2, dim=1) \
            - 2 * torch.einsum("bd,dn->bn", flat, e.t())  # quantize.py:280-282
        idx = torch.argmin(d, dim=1)  # quantize.py:284
        zq = self.embedding(idx).view(zp.shape)
        if not self.legacy:  # quantize.py:290-295
            loss = self.beta * torch.mean((zq.detach() - zp) ** 2) + torch.mean((zq - zp.detach()) ** 2)
        else:
            loss = torch.mean((zq.detach() - zp) ** 2) + self.beta * torch.mean((zq - zp.detach()) ** 2)
        zq = zp + (zq - zp).detach()  # quantize.py:298
        return zq.permute(0, 3, 1, 2).contiguous(), loss, (None, None, idx)


class VQModel(nn.Module):  # VQModel_arch.py:14-91 (encode/decode only)
    def __init__(self, resolution=256, n_embed=8192, embed_dim=3, z_channels=3, in_channels=3, out_ch=3,
                 ch=128, ch_mult=(1, 2, 4), num_res_blocks=2, attn_resolutions=(64,)):
        super().__init__()
        self.encoder = Encoder(ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, in_channels,
                               resolution, z_channels, double_z=False)
        self.decoder = Decoder(ch, out_ch, ch_mult, num_res_blocks, attn_resolutions, resolution, z_channels)
        self.quantize = VectorQuantizer2(n_embed, embed_dim, beta=0.25)
        self.quant_conv = nn.Conv2d(z_channels, embed_dim, 1)
        self.post_quant_conv = nn.Conv2d(embed_dim, z_channels, 1)
        self.conv_semantic = nn.Sequential(nn.Conv2d(3, 256, 1, 1, 0), nn.ReLU())  # unused on the path

    def encode(self, x):  # VQModel_arch.py:74-79
        return self.quant_conv(self.encoder(x)), None

    def decode(self, h):  # VQModel_arch.py:81-91
        quant, emb_loss, info = self.quantize(h)
        dec, feats = self.decoder(self.post_quant_conv(quant))
        self.last_indices = info[2]
        return dec, emb_loss, feats


# --------------------------------------------------------------------------------------------
# Conditional encoder                                        ConditionEncoder.py:14-55
# --------------------------------------------------------------------------------------------
class ConEncoder1(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoder = Encoder(128, 3, (1, 2, 4), 2, (64,), 3, 256, 3, False)
        self.color_conv = nn.Conv2d(3, 3, 3, 1, 1)
        self.cond_conv = nn.Sequential(nn.Conv2d(3, 64, 3, 1, 1), nn.Sigmoid())

    def forward(self, x, mid_feat=False):
        enc, feats = self.encoder(x, mid_feat=True)
        out = {"cond_feat": self.cond_conv(enc), "color_map": self.color_conv(enc)}
        if mid_feat:
            out["mid_feat"] = feats
        return out


# --------------------------------------------------------------------------------------------
# Normalizing flow              FlowActNorms.py, Permutations.py, flow.py, FlowStep.py, ...
# --------------------------------------------------------------------------------------------
class ActNorm2d(nn.Module):  # FlowActNorms.py:10-100
    def __init__(self, c, scale=1.0):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(1, c, 1, 1))
        self.logs = nn.Parameter(torch.zeros(1, c, 1, 1))
        self.scale = float(scale)
        self.inited = False  # :23

    def initialize_parameters(self, x):  # :32-46: data-dependent init on the first TRAINING forward of an all-zero bias
        if not self.training:
            return
        if (self.bias != 0).any():
            self.inited = True
            return
        with torch.no_grad():
            def mean(t):  # thops.mean (thops.py:19-33): one dimension at a time, in sorted order
                for d in (0, 2, 3):
                    t = t.mean(dim=d, keepdim=True)
                return t

            bias = mean(x.clone()) * -1.0
            var = mean((x.clone() + bias) ** 2)
            logs = torch.log(self.scale / (torch.sqrt(var) + 1e-6))
            self.bias.data.copy_(bias.data)
            self.logs.data.copy_(logs.data)
            self.inited = True

    def forward(self, x, logdet=None, reverse=False):
        if not self.inited:  # :82-83
            self.initialize_parameters(x)
        pixels = x.shape[2] * x.shape[3]
        if not reverse:
            x = (x + self.bias) * torch.exp(self.logs)
            d = self.logs.sum() * pixels
        else:
            x = x * torch.exp(-self.logs) - self.bias
            d = -self.logs.sum() * pixels
        if logdet is not None:
            logdet = logdet + d
        return x, logdet


class InvertibleConv1x1(nn.Module):  # Permutations.py:12-59
    def __init__(self, c):
        super().__init__()
        w = np.linalg.qr(np.random.randn(c, c))[0].astype(np.float32)
        self.weight = nn.Parameter(torch.from_numpy(w))

    def forward(self, x, logdet=None, reverse=False):
        c = self.weight.shape[0]
        pixels = x.shape[2] * x.shape[3]
        d = torch.slogdet(self.weight)[1] * pixels
        if not reverse:
            w = self.weight.view(c, c, 1, 1)
        else:
            w = torch.inverse(self.weight.double()).float().view(c, c, 1, 1)  # Permutations.py:38
            d = -d
        z = F.conv2d(x, w)
        if logdet is not None:
            logdet = logdet + d
        return z, logdet


class FlowConv2d(nn.Conv2d):  # flow.py:13-52 ("Conv2d": bias-free conv followed by ActNorm2d)
    def __init__(self, cin, cout, k=3):
        super().__init__(cin, cout, k, 1, (k - 1) // 2, bias=False)
        self.weight.data.normal_(0.0, 0.05)
        self.actnorm = ActNorm2d(cout)

    def forward(self, x):
        return self.actnorm(super().forward(x))[0]


class Conv2dZeros(nn.Conv2d):  # flow.py:55-70
    def __init__(self, cin, cout):
        super().__init__(cin, cout, 3, 1, 1)
        self.logs = nn.Parameter(torch.zeros(cout, 1, 1))
        self.weight.data.zero_()
        self.bias.data.zero_()

    def forward(self, x):
        return super().forward(x) * torch.exp(self.logs * 3)


def _coupling_net(cin, cout, hidden=64):  # FlowAffineCouplingsAblation.py:143-151
    return nn.Sequential(FlowConv2d(cin, hidden, 3), nn.ReLU(), FlowConv2d(hidden, hidden, 1), nn.ReLU(),
                         Conv2dZeros(hidden, cout))


class CondAffineSeparatedAndCond(nn.Module):  # FlowAffineCouplingsAblation.py:10-151
    def __init__(self, c=3, c_cond=64):
        super().__init__()
        self.c_nn = c // 2
        self.c_co = c - self.c_nn
        self.eps = 0.0001
        self.fAffine = _coupling_net(self.c_nn + c_cond, self.c_co * 2)
        self.fFeatures = _coupling_net(c_cond, c * 2)

    def _scale_shift(self, h):  # :124-135 ("cross" split, thops.py:39-47)
        shift, scale = h[:, 0::2], h[:, 1::2]
        return torch.sigmoid(scale + 2.0) + self.eps, shift

    def forward(self, z, logdet, reverse, ft):
        if not reverse:  # :51-81
            s_ft, t_ft = self._scale_shift(self.fFeatures(ft))
            z = (z + t_ft) * s_ft
            logdet = logdet + torch.log(s_ft).sum(dim=[1, 2, 3])
            z1, z2 = z[:, :self.c_nn], z[:, self.c_nn:]
            s, t = self._scale_shift(self.fAffine(torch.cat([z1, ft], 1)))
            z2 = (z2 + t) * s
            logdet = logdet + torch.log(s).sum(dim=[1, 2, 3])
            z = torch.cat([z1, z2], 1)
        else:  # :83-110
            z1, z2 = z[:, :self.c_nn], z[:, self.c_nn:]
            s, t = self._scale_shift(self.fAffine(torch.cat([z1, ft], 1)))
            z2 = z2 / s - t
            z = torch.cat([z1, z2], 1)
            logdet = logdet - torch.log(s).sum(dim=[1, 2, 3])
            s_ft, t_ft = self._scale_shift(self.fFeatures(ft))
            z = z / s_ft - t_ft
            logdet = logdet - torch.log(s_ft).sum(dim=[1, 2, 3])
        return z, logdet


class FlowStep(nn.Module):  # FlowStep.py:16-127
    def __init__(self, c=3, coupling=True):
        super().__init__()
        self.actnorm = ActNorm2d(c)
        self.invconv = InvertibleConv1x1(c)
        if coupling:
            self.affine = CondAffineSeparatedAndCond(c)
        self.coupling = coupling

    def forward(self, z, logdet, reverse, ft):
        if not reverse:  # :75-98
            z, logdet = self.actnorm(z, logdet, False)
            z, logdet = self.invconv(z, logdet, False)
            if self.coupling:
                z, logdet = self.affine(z, logdet, False, ft)
        else:  # :100-119
            if self.coupling:
                z, logdet = self.affine(z, logdet, True, ft)
            z, logdet = self.invconv(z, logdet, True)
            z, logdet = self.actnorm(z, logdet, True)
        return z, logdet


class FlowUpsamplerNet(nn.Module):  # FlowUpsamplerNet.py:17-326 at confs/LOL.yml (L=2, K=12, 2 extra)
    def __init__(self, L=2, K=12, n_extra=2, c=3):
        super().__init__()
        self.layers = nn.ModuleList()
        for _ in range(L):
            for _ in range(n_extra):
                self.layers.append(FlowStep(c, coupling=False))
            for _ in range(K):
                self.layers.append(FlowStep(c, coupling=True))
        # built but never called (FlowUpsamplerNet.py:113-116): 320 -> 384 3x3 conv
        self.f = nn.Sequential(nn.Conv2d(320, 384, 3, 1, 1))

    def encode(self, gt, ft, logdet):  # :228-274
        z = gt
        for layer in self.layers:
            z, logdet = layer(z, logdet, False, ft)
        return z, logdet

    def forward(self, gt=None, rrdbResults=None, z=None, epses=None, logdet=0.0, reverse=False, eps_std=None, y_onehot=None):
        """The reference's call form (FlowUpsamplerNet.py:216-226)."""
        ft = rrdbResults["cond_feat"] if isinstance(rrdbResults, dict) else rrdbResults
        return self.decode(z, ft, logdet) if reverse else self.encode(gt, ft, logdet)

    def decode(self, z, ft, logdet=None):  # :290-326
        if logdet is None:
            logdet = torch.zeros_like(z[:, 0, 0, 0])
        for layer in reversed(self.layers):
            z, logdet = layer(z, logdet, True, ft)
        return z, logdet


# --------------------------------------------------------------------------------------------
# DCNv2                      ops/dcn/deform_conv.py + ops/dcn/src/deform_conv_cuda{.cpp,_kernel.cu}
# --------------------------------------------------------------------------------------------
def modulated_deform_conv(x, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                          deformable_groups=1):
    """Differentiable pure-torch DCNv2 (autograd supplies all five gradients).

    col[b, c, k, h, w] = mask * bilinear(x[b, c], h*s - p + i*d + dh, w*s - p + j*d + dw)
        (deform_conv_cuda_kernel.cu:571-633; corner handling :468-497; validity test :618)
    out = W.flatten(1) @ col + bias            (deform_conv_cuda.cpp:539-568)
    offset channel of (group g, tap k): g*2K + 2k (dh), +1 (dw); mask channel g*K + k.
    """
    B, C, H, W_ = x.shape
    Co, Cg, kh, kw = weight.shape
    K = kh * kw
    dg = deformable_groups
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W_ + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    cpg = C // dg
    dev, dt = x.device, x.dtype
    hs = (torch.arange(Ho, device=dev, dtype=dt) * stride - padding).view(1, 1, Ho, 1)
    ws = (torch.arange(Wo, device=dev, dtype=dt) * stride - padding).view(1, 1, 1, Wo)
    off = offset.view(B, dg, K, 2, Ho, Wo)
    msk = mask.view(B, dg, K, Ho, Wo)
    xg = x.reshape(B, dg, cpg, H * W_)
    cols = []
    for k in range(K):
        i, j = k // kw, k % kw
        h_im = hs + i * dilation + off[:, :, k, 0]  # [B, dg, Ho, Wo]
        w_im = ws + j * dilation + off[:, :, k, 1]
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W_)
        h_low, w_low = torch.floor(h_im), torch.floor(w_im)
        lh, lw = h_im - h_low, w_im - w_low
        hh, hw = 1 - lh, 1 - lw
        h_low, w_low = h_low.long(), w_low.long()
        h_high, w_high = h_low + 1, w_low + 1

        def corner(hi, wi, ok):
            ok = ok & inside
            lin = (hi.clamp(0, H - 1) * W_ + wi.clamp(0, W_ - 1)).view(B, dg, 1, Ho * Wo).expand(B, dg, cpg, Ho * Wo)
            v = torch.gather(xg, 3, lin).view(B, dg, cpg, Ho, Wo)
            return v * ok.unsqueeze(2).to(dt)

        v1 = corner(h_low, w_low, (h_low >= 0) & (w_low >= 0))
        v2 = corner(h_low, w_high, (h_low >= 0) & (w_high <= W_ - 1))
        v3 = corner(h_high, w_low, (h_high <= H - 1) & (w_low >= 0))
        v4 = corner(h_high, w_high, (h_high <= H - 1) & (w_high <= W_ - 1))
        w1, w2, w3, w4 = (hh * hw).unsqueeze(2), (hh * lw).unsqueeze(2), (lh * hw).unsqueeze(2), (lh * lw).unsqueeze(2)
        val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
        cols.append(val * msk[:, :, k].unsqueeze(2))
    col = torch.stack(cols, dim=3).reshape(B, groups, (C // groups) * K, Ho * Wo)  # channel-major, tap-minor
    wmat = weight.reshape(groups, Co // groups, Cg * K)
    out = torch.einsum("gok,bgkn->bgon", wmat, col).reshape(B, Co, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, Co, 1, 1)
    return out


class DCNv2Pack(nn.Module):  # deformableDecoder_arch.py:132-152 over deform_conv.py:289-379
    def __init__(self, cin, cout, k=3, padding=1, deformable_groups=4):
        super().__init__()
        self.stride, self.padding, self.dilation, self.groups = 1, padding, 1, 1
        self.deformable_groups = deformable_groups
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        self.bias = nn.Parameter(torch.zeros(cout))
        stdv = 1.0 / math.sqrt(cin * k * k)
        self.weight.data.uniform_(-stdv, stdv)
        self.conv_offset = nn.Conv2d(cin, deformable_groups * 3 * k * k, k, 1, padding)
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def offsets(self, feat):
        out = self.conv_offset(feat).float()
        o1, o2, m = torch.chunk(out, 3, dim=1)
        return torch.cat((o1, o2), dim=1), torch.sigmoid(m)

    def forward(self, x, feat):
        offset, mask = self.offsets(feat)
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                     self.dilation, self.groups, self.deformable_groups)


class WarpBlock(nn.Module):  # deformableDecoder_arch.py:279-290
    def __init__(self, c):
        super().__init__()
        self.offset = nn.Conv2d(c * 2, c, 3, 1, 1)
        self.dcn = DCNv2Pack(c, c, 3, padding=1, deformable_groups=4)

    def forward(self, x_vq, x_res):
        return self.dcn(x_vq, self.offset(torch.cat([x_vq, x_res], dim=1)))


class Mix(nn.Module):  # deformableDecoder_arch.py:579-590
    def __init__(self, m):
        super().__init__()
        self.w = nn.Parameter(torch.FloatTensor([m]))

    def forward(self, a, b):
        f = torch.sigmoid(self.w)
        return a * f.expand_as(a) + b * (1 - f.expand_as(b))


class _UnusedResBlock(nn.Module):  # deformableDecoder_arch.py:157-180 (parameters only)
    def __init__(self, cin, cout):
        super().__init__()
        self.norm1 = group_norm(cin)
        self.conv1 = nn.Conv2d(cin, cout, 3, 1, 1)
        self.norm2 = group_norm(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1)
        self.conv_out = nn.Conv2d(cin, cout, 1, 1, 0)


def _unused_gate(c):  # deformableDecoder_arch.py:490-508 (parameters only)
    return nn.Sequential(nn.Conv2d(c, c, 3, padding=1), nn.LeakyReLU(0.2, True), nn.Conv2d(c, c, 3, padding=1),
                         nn.Sigmoid())


class MultiScaleDecoder2(nn.Module):  # deformableDecoder_arch.py:413-576
    def __init__(self, ch=128, out_ch=3, ch_mult=(1, 2, 4), num_res_blocks=2, attn_resolutions=(64,),
                 resolution=256, z_channels=3, per_sample_mean=False):
        super().__init__()
        cin = _decoder_trunk(self, ch, ch_mult, num_res_blocks, attn_resolutions, resolution, z_channels)
        self.conv_out = nn.Conv2d(cin, out_ch, 3, 1, 1)  # unused by forward
        self.warp = nn.ModuleList([WarpBlock(ch * 2), WarpBlock(ch)])
        self.residual_conv = nn.Conv2d(ch, out_ch, 3, 1, 1)
        self.scale = nn.ModuleList([_unused_gate(256), _unused_gate(128)])
        self.bias = nn.ModuleList([_unused_gate(256), _unused_gate(128)])
        self.enc = nn.ModuleList([_UnusedResBlock(512, 256), _UnusedResBlock(256, 128)])
        self.mix = nn.ModuleList([Mix(-1.0), Mix(-0.6)])
        # False = the reference's whole-batch means (:567); True = per-sample means, the build's
        # inference contract -- identical at B = 1.
        self.per_sample_mean = per_sample_mean

    def forward(self, z, code_feats, enc_feats):
        h = self.conv_in(z)
        h = self.mid.block_2(self.mid.attn_1(self.mid.block_1(h)))
        for lvl in reversed(range(self.num_resolutions)):
            level = self.up[lvl]
            for i in range(self.num_res_blocks + 1):
                h = level.block[i](h)
                if len(level.attn) > 0:
                    h = level.attn[i](h)
            if lvl != 2:  # :546-567
                x_code = code_feats[1 - lvl].float()
                h = self.mix[1 - lvl](enc_feats[lvl], h.float())
                x_w = self.warp[1 - lvl](x_code, h)
                if self.per_sample_mean:
                    ratio = h.mean(dim=[1, 2, 3], keepdim=True) / x_w.mean(dim=[1, 2, 3], keepdim=True)
                else:
                    ratio = h.mean() / x_w.mean()
                h = h + x_w * ratio
            if lvl != 0:
                h = level.upsample(h)
        return self.residual_conv(swish(self.norm_out(h)))


# --------------------------------------------------------------------------------------------
# Graphs                            VQLLFLOWDeformable_arch.py, LLFlowVQGAN_arch.py
# --------------------------------------------------------------------------------------------
class VQLLFLOWDeformable(nn.Module):  # VQLLFLOWDeformable_arch.py:18-250 (reverse path)
    def __init__(self, per_sample_mean=False):
        super().__init__()
        self.RRDB = ConEncoder1()
        self.deformable_decoder = MultiScaleDecoder2(per_sample_mean=per_sample_mean)
        self.flowUpsamplerNet = FlowUpsamplerNet()

    def stages(self, net_vq, lr):
        """reverse_flow (:222-250) returning every stage boundary for parity checks."""
        enc = self.RRDB(lr, mid_feat=True)
        x, _ = self.flowUpsamplerNet.decode(enc["color_map"], enc["cond_feat"])
        rec, _, code_feats = net_vq.decode(x)
        out = self.deformable_decoder(x, list(code_feats), enc["mid_feat"])
        return {"enc": enc, "latent": x, "indices": net_vq.last_indices, "code_feats": code_feats, "vq_rec": rec,
                "out": out}

    def forward(self, net_vq, lr):
        s = self.stages(net_vq, lr)
        return s["out"], s["latent"]


class LLFlowVQGAN2(nn.Module):  # LLFlowVQGAN_arch.py:17-106 (stage 2, normal flow)
    def __init__(self):
        super().__init__()
        self.RRDB = ConEncoder1()
        self.flowUpsamplerNet = FlowUpsamplerNet()

    def normal_flow(self, gt, lr, train_gt_ratio=0.0):
        """train_gt_ratio: confs/LOL.yml:12 = 0, confs/train_stage2_LOL.yml:14 = 0.2; one `random.random()` draw per call
        decides whether the Gaussian's mean is color_map or the ground-truth latent (LLFlowVQGAN_arch.py:95)."""
        import random

        enc = self.RRDB(lr)
        pixels = gt.shape[2] * gt.shape[3]
        logdet = torch.zeros_like(gt[:, 0, 0, 0])
        z, logdet = self.flowUpsamplerNet.encode(gt, enc["cond_feat"], logdet)
        mean = enc["color_map"] if random.random() > train_gt_ratio else gt  # LLFlowVQGAN_arch.py:95
        logp = (-0.5 * ((z - mean) ** 2 + float(np.log(2 * np.pi)))).sum(dim=[1, 2, 3])  # flow.py:76-95
        nll = -(logdet + logp) / float(np.log(2.0) * pixels)  # LLFlowVQGAN_arch.py:99-101
        return z, nll, logdet


# --------------------------------------------------------------------------------------------
# Harness pre/post-processing                      infer_dataset_lol.py:113-153, utils2.py:32-36
# --------------------------------------------------------------------------------------------
def preprocess(img_u8):
    """uint8 HxWx3 -> 1x3x(H+20)x(W+20) fp32 log-domain (infer_dataset_lol.py:124-128,42,71-72)."""
    img = np.pad(img_u8, [(0, 20), (20, 0), (0, 0)], "reflect")
    t = torch.Tensor(np.expand_dims(img.transpose([2, 0, 1]), axis=0).astype(np.float32)) / 255
    return torch.log(torch.clamp(t + 1e-3, min=1e-3))


def postprocess(out, h, gt_u8=None):
    """crop, clamp, optional GT-mean gain (infer_dataset_lol.py:135-144); returns HxWx3 float."""
    r = torch.clamp(out[:, :, :h, 20:], 0, 1).permute(0, 2, 3, 1).squeeze(0).numpy()
    if gt_u8 is not None:
        tgt = gt_u8 / 255
        r = np.clip(r * (gray_mean(tgt) / gray_mean(r)), 0, 1)
    return r


def gray_mean(img):
    """cv2.COLOR_BGR2GRAY weights applied to RGB-ordered data (infer_dataset_lol.py:142-143)."""
    img = img.astype(np.float32)
    return (0.114 * img[..., 0] + 0.587 * img[..., 1] + 0.299 * img[..., 2]).mean()


def ssim_utils2(img1_u8, img2_u8):
    """calculate_ssim (code/utils/utils2.py:42-89) on HxWx3 uint8 images, float64: per channel, the 11x11 Gaussian window
    (cv2.getGaussianKernel(11, 1.5) = normalised exp(-(i-5)^2 / (2 * 1.5^2)), cv2's documented formula) at the VALID positions
    (filter2D(...)[5:-5, 5:-5]), C1 = (0.01*255)^2, C2 = (0.03*255)^2, mean of the map; channels averaged.
    PARITY UNPINNED BY EXECUTION (cv2 is not in this image); pinned instead to the reference's own pytorch_msssim.ssim, which is
    the same formula and IS executable here (tests/golden/ssim_metric.npz), and by SSIM(x, x) = 1."""
    k = np.exp(-((np.arange(11) - 5.0) ** 2) / (2 * 1.5 ** 2))
    k = k / k.sum()
    C1, C2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2

    def filt(a):   # valid separable correlation
        a = np.stack([np.convolve(r, k, mode="valid") for r in a])                 # along x
        return np.stack([np.convolve(c, k, mode="valid") for c in a.T]).T          # along y

    vals = []
    for c in range(3):
        a, b = img1_u8[:, :, c].astype(np.float64), img2_u8[:, :, c].astype(np.float64)
        mu1, mu2 = filt(a), filt(b)
        s1, s2, s12 = filt(a * a) - mu1 ** 2, filt(b * b) - mu2 ** 2, filt(a * b) - mu1 * mu2
        vals.append((((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 ** 2 + mu2 ** 2 + C1) * (s1 + s2 + C2))).mean())
    return float(np.mean(vals))


def psnr(a, b):  # utils2.py:32-36
    mse = np.mean((a - b) ** 2)
    return 10 * np.log10(1.0 / mse)

# --------------------------------------------------------------------------------------------
# LPIPS (AlexNet)            Measure.py:17-30 -> third-party package `lpips` (absent from /root/reference)
# --------------------------------------------------------------------------------------------
class LPIPSAlex(nn.Module):
    """`lpips.LPIPS(net='alex')` (Measure.py:20) restated from the package's published definition (Zhang et al., CVPR 2018; PyPI
    lpips 0.1.x, `lpips/lpips.py` + `lpips/pretrained_networks.py`; the reference does not pin a version).  PARITY UNPINNED BY
    EXECUTION: neither the package, nor torchvision's AlexNet, nor their downloaded weights exist in this image; anchored on the
    reference's call site (uint8 HWC images -> Measure.t -> x / 127.5 - 1 -> model.forward(tA, tB).item()) and on the metric's
    properties (d(x, x) = 0, symmetry, non-negativity for non-negative heads).
      forward(in0, in1): ScalingLayer (x - shift) / scale; alexnet.features cut at the five ReLUs (conv 11x11 s4 p2 | maxpool 3 s2,
      conv 5x5 p2 | maxpool 3 s2, conv 3x3 p1 | conv 3x3 p1 | conv 3x3 p1); per tap k: normalize_tensor(f) = f / (sqrt(sum_c f^2) + 1e-10),
      d = (n0 - n1)^2, lin_k = 1x1 conv C_k -> 1 without bias (Dropout in front is the identity in eval), spatial mean; sum over k.
    Same state-dict keys as the package (and as glare_amd.metrics.LPIPS)."""

    CONVS = ((0, 3, 64, 11, 4, 2), (3, 64, 192, 5, 1, 2), (6, 192, 384, 3, 1, 1), (8, 384, 256, 3, 1, 1), (10, 256, 256, 3, 1, 1))

    def __init__(self):
        super().__init__()
        self.scaling_layer = nn.Module()
        self.scaling_layer.register_buffer("shift", torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.scaling_layer.register_buffer("scale", torch.Tensor([.458, .448, .450])[None, :, None, None])
        self.net = nn.Module()
        for s, (idx, cin, cout, k, st, pd) in enumerate(self.CONVS):
            seq = nn.Sequential()
            if s in (1, 2):
                seq.add_module(str(idx - 1), nn.MaxPool2d(kernel_size=3, stride=2))
            seq.add_module(str(idx), nn.Conv2d(cin, cout, kernel_size=k, stride=st, padding=pd))
            seq.add_module(str(idx + 1), nn.ReLU(inplace=False))
            setattr(self.net, "slice%d" % (s + 1), seq)
        for k, (_, _, c, _, _, _) in enumerate(self.CONVS):
            lin = nn.Module()
            lin.model = nn.Sequential(nn.Dropout(), nn.Conv2d(c, 1, 1, stride=1, padding=0, bias=False))
            setattr(self, "lin%d" % k, lin)
        self.lins = nn.ModuleList([getattr(self, "lin%d" % k) for k in range(5)])
        self.eval()

    def forward(self, in0, in1, normalize=False):
        if normalize:
            in0, in1 = 2 * in0 - 1, 2 * in1 - 1
        h0 = (in0 - self.scaling_layer.shift) / self.scaling_layer.scale
        h1 = (in1 - self.scaling_layer.shift) / self.scaling_layer.scale
        val = 0
        for k in range(5):
            sl = getattr(self.net, "slice%d" % (k + 1))
            h0, h1 = sl(h0), sl(h1)
            n0 = h0 / (torch.sqrt(torch.sum(h0 ** 2, dim=1, keepdim=True)) + 1e-10)
            n1 = h1 / (torch.sqrt(torch.sum(h1 ** 2, dim=1, keepdim=True)) + 1e-10)
            val = val + self.lins[k].model(((n0 - n1) ** 2)).mean([2, 3], keepdim=True)
        return val


def lpips_input(img_u8):
    """Measure.t (Measure.py:48-64): HWC uint8 -> 1x3xHxW float in [-1, 1]."""
    return torch.Tensor(np.transpose(img_u8, [2, 0, 1])[None]) / 127.5 - 1



# --------------------------------------------------------------------------------------------
# Stage-3 loss terms         modules/pytorch_msssim/__init__.py, modules/losses.py, VQLLFLOWD_model.py:209-223
# --------------------------------------------------------------------------------------------
def _gaussian(window_size, sigma):  # pytorch_msssim/__init__.py:8-10
    from math import exp
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def _create_window(window_size, channel=1):  # :13-17
    w1 = _gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def ssim(img1, img2, window_size=11, val_range=None):
    """ssim(..., full=True) of :21-68: returns (mean ssim_map, mean cs_map); NCHW."""
    if val_range is None:
        max_val = 255 if torch.max(img1) > 128 else 1
        min_val = -1 if torch.min(img1) < -0.5 else 0
        L = max_val - min_val
    else:
        L = val_range
    _, channel, height, width = img1.size()
    window = _create_window(min(window_size, height, width), channel=channel).to(img1.device)
    mu1 = F.conv2d(img1, window, padding=0, groups=channel)
    mu2 = F.conv2d(img2, window, padding=0, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=0, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=0, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=0, groups=channel) - mu1_mu2
    C1, C2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    v1, v2 = 2.0 * sigma12 + C2, sigma1_sq + sigma2_sq + C2
    cs = torch.mean(v1 / v2)
    ssim_map = ((2 * mu1_mu2 + C1) * v1) / ((mu1_sq + mu2_sq + C1) * v2)
    return ssim_map.mean(), cs


def msssim(img1, img2, window_size=11, val_range=None, normalize=False):  # :71-98
    weights = torch.FloatTensor([0.0448, 0.2856, 0.3001, 0.2363, 0.1333]).to(img1.device)
    mssim, mcs = [], []
    for _ in range(weights.size()[0]):
        sim, cs = ssim(img1, img2, window_size=window_size, val_range=val_range)
        mssim.append(sim)
        mcs.append(cs)
        img1, img2 = F.avg_pool2d(img1, (2, 2)), F.avg_pool2d(img2, (2, 2))
    mssim, mcs = torch.stack(mssim), torch.stack(mcs)
    if normalize:
        mssim, mcs = (mssim + 1) / 2, (mcs + 1) / 2
    pow1, pow2 = mcs ** weights, mssim ** weights
    return torch.prod(pow1[:-1] * pow2[-1])


class PerceptualNetwork(nn.Module):
    """losses.py:12-40 with vgg16.features[:16] written out (torchvision cfg 'D': 64 64 M 128 128 M 256 256 256).  The
    reference loads torchvision's pretrained weights; torchvision is absent here, so the structure is restated from its
    published definition and the weights are the caller's: parity unpinned by execution for this module."""

    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in (64, 64, "M", 128, 128, "M", 256, 256, 256):
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=False)]
                cin = v
        self.vgg_model = nn.Sequential(*layers)
        for p in self.vgg_model.parameters():
            p.requires_grad = False

    def output_features(self, x):
        out = []
        for i, m in enumerate(self.vgg_model):
            x = m(x)
            if i in (3, 8, 15):
                out.append(x)
        return out

    def forward(self, dehaze, gt):
        fa, fb = self.output_features(dehaze), self.output_features(gt)
        return sum(F.mse_loss(a, b) for a, b in zip(fa, fb)) / len(fa)


def stage3_loss(rec, real_H, perceptual):
    """VQLLFLOWD_model.py:209-223: returns (total, l1, percep * 0.01, ssim * 0.2)."""
    sr = rec.to(torch.float32).clamp(0, 1)
    not_nan = ~torch.isnan(sr)
    sr = torch.where(not_nan, sr, torch.zeros_like(sr))
    l1 = ((sr - real_H) * not_nan).abs().mean()
    pl = perceptual(sr, real_H) * 0.01
    sl = (1 - msssim(sr, real_H, normalize=True)) * 0.2
    return l1 + pl + sl, l1, pl, sl
