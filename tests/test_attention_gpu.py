"""GPU: the fused attention forward + VJP (csrc/attention.hip) through the C-ABI test hook against a plain PyTorch fp32
restatement of QKVAttentionLegacy (guided_diffusion/unet.py:339-356) and its autograd gradient on the same bf16-rounded inputs.
Tolerances are written next to each check: probabilities, dS and all outputs pass through bf16 (2^-9 relative rounding)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def legacy_attention(qkv, heads):
    """qkv [B, T, 3C] with head h at channels 192 h + (q | k | v) -> [B, T, C] (fp32 throughout)."""
    B, T, W3 = qkv.shape
    ch = W3 // (3 * heads)
    x = qkv.view(B, T, heads, 3, ch)
    q, k, v = x[:, :, :, 0], x[:, :, :, 1], x[:, :, :, 2]                # [B, T, heads, ch]
    scale = ch ** -0.25
    w = torch.einsum("bthc,bshc->bhts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bhts,bshc->bthc", w, v)
    return a.reshape(B, T, heads * ch)


@pytest.mark.parametrize("B,T,heads", [(2, 64, 2), (1, 256, 4), (1, 1024, 1)])
def test_fused_attention_forward_and_vjp(B, T, heads):
    import kdip_amd._lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(7)
    Cc = 64 * heads
    qkv = torch.randn(B, T, 3 * Cc, generator=g) * 1.5           # scores with a spread of a few units: a peaked softmax
    dO = torch.randn(B, T, Cc, generator=g)
    qd, dd = qkv.cuda().contiguous(), dO.cuda().contiguous()
    o = torch.empty(B, T, Cc, device="cuda"); dq = torch.empty(B, T, 3 * Cc, device="cuda")
    L.check(lib.kdip_test_attention(L.stream(), L.ptr(qd), L.ptr(dd), B, T, heads, L.ptr(o), L.ptr(dq)))
    torch.cuda.synchronize()
    qr = bf(qkv).requires_grad_()
    ref = legacy_attention(qr, heads)
    gref = torch.autograd.grad((ref * bf(dO)).sum(), qr)[0]
    e_o = float((o.cpu() - ref.detach()).abs().max() / ref.detach().abs().max())
    e_g = float((dq.cpu() - gref).abs().max() / gref.abs().max())
    print(f"\nfused attention B={B} T={T} heads={heads}: forward rel err {e_o:.2e}, VJP rel err {e_g:.2e}")
    assert e_o < 1.5e-2        # bf16 probabilities (unnormalised, <= 1) and bf16 output
    assert e_g < 3e-2          # + bf16 dS and the bf16-rounded saved output in D = rowsum(dO * O)
