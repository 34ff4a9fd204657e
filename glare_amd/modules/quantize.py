"""Codebook retrieval on the HIP nearest-code kernel.

Mirrors VectorQuantizer2 (code/models/modules/quantize.py:213-329): same constructor, same
`forward(z) -> (z_q, loss, (perplexity, min_encodings, min_encoding_indices))` contract with
`sane_index_shape=False` semantics (flat int64 indices), `legacy=False` loss, straight-through z_q."""
import torch
import torch.nn as nn

from .. import ops
from ._base import HipModule


class VectorQuantizer2(HipModule):
    def __init__(self, n_e, e_dim, beta, remap=None, unknown_index="random", sane_index_shape=False, legacy=False):
        super().__init__()
        assert remap is None, "index remapping is not used by GLARE (VQModel_arch.py:44-45)"
        self.n_e, self.e_dim, self.beta, self.legacy = n_e, e_dim, beta, legacy
        self.embedding = nn.Embedding(n_e, e_dim)
        self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)
        self.remap, self.re_embed, self.sane_index_shape = None, n_e, sane_index_shape

    def quantize_tokens(self, tokens):
        """tokens fp32 [N, e_dim] (NHWC flattening) -> (idx int64 [N], z_q fp32 [N, e_dim])."""
        return ops.vq_nearest(tokens, self.embedding.weight.detach())

    def forward(self, z, temp=None, rescale_logits=False, return_logits=False):
        assert temp is None or temp == 1.0
        assert not rescale_logits and not return_logits
        ops.require_cuda(z)
        if torch.is_grad_enabled() and z.requires_grad:
            # GLARE only ever runs the codebook under no_grad (VQLLFLOWDeformable_arch.py:245, LLFlow_model.py:200): the HIP
            # lookup has no tape, so the commitment / codebook losses below could not train anything -- refuse loudly
            raise NotImplementedError("VectorQuantizer2 on HIP is inference-only (stage-1 VQGAN training is out of scope): "
                                      "call it under torch.no_grad()")
        zp = ops.nchw_to_nhwc(z, bf16=False)  # 'b c h w -> b h w c' (quantize.py:276)
        flat = zp.view(-1, self.e_dim)
        idx, zq = self.quantize_tokens(flat)
        zq = zq.view(zp.shape)
        # losses and the straight-through estimator are cheap elementwise torch ops on device (quantize.py:290-298)
        if not self.legacy:
            loss = self.beta * torch.mean((zq.detach() - zp) ** 2) + torch.mean((zq - zp.detach()) ** 2)
        else:
            loss = torch.mean((zq.detach() - zp) ** 2) + self.beta * torch.mean((zq - zp.detach()) ** 2)
        zq = zp + (zq - zp).detach()
        zq = ops.nhwc_to_nchw(zq.contiguous())
        if self.sane_index_shape:
            idx = idx.reshape(zq.shape[0], zq.shape[2], zq.shape[3])
        return zq, loss, (None, None, idx)

    def get_codebook_entry(self, indices, shape):
        zq = self.embedding(indices)
        if shape is not None:
            zq = zq.view(shape).permute(0, 3, 1, 2).contiguous()
        return zq
