"""Reads the ATTN_PROFILE counters (build with GLARE_DEFS=-DATTN_PROFILE): issue-time per phase, cycles per tile."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glare_amd import ops
B, N, C = 8, 105 * 155, 512
qk = (torch.randn(B, N, 2 * C, device="cuda") * 0.3).to(torch.bfloat16)
npad = (N + 63) // 64 * 64
vt = torch.zeros(B, C, npad, dtype=torch.bfloat16, device="cuda")
vt[:, :, :N] = torch.randn(B, C, N, device="cuda").to(torch.bfloat16)
full = torch.zeros(B * N * C + 64, dtype=torch.bfloat16, device="cuda")
out = full[:B * N * C].view(B, N, C)
ops.attention_d512(qk, qk[..., C:], vt, N, ldq=2 * C, ldk=2 * C, out=out)
torch.cuda.synchronize()
raw = full[B * N * C:B * N * C + 32].view(torch.int64).cpu().tolist()
nt = raw[6]
names = ["wait+barrier", "QK^T phase", "softmax", "PV phase", "-", "loop"]
tot = sum(raw[:6])
for n, v in zip(names, raw[:6]):
    print("%-14s %8.0f ticks/tile" % (n, v / nt))
print("total %.0f ticks/tile (s_memtime ticks = 100 MHz? see ratio), tiles %d" % (tot / nt, nt))
