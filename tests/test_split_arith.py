"""The arithmetic the bf16x6 kernel (conv_gemm_split.hip K17) rests on, restated with torch on the CPU: the nearest-rounding three-way split of an fp32 value is
EXACT, every piece is at most half a bf16 ulp of the one before, piece products are exact in fp32, and the six products of order <= 2 reproduce the fp32
product to ~2^-24 (the three dropped ones are ~2^-25).  tests/precision/split_probe.py runs whole networks through the same emulation."""
import numpy as np
import torch


def pieces(v):
    p0 = v.bfloat16().float()
    r1 = v - p0
    p1 = r1.bfloat16().float()
    r2 = r1 - p1
    p2 = r2.bfloat16().float()
    return p0, p1, p2, r2


def test_three_bf16_pieces_are_exact():
    g = torch.Generator().manual_seed(0)
    v = torch.randn(200000, generator=g) * torch.exp(12 * torch.randn(200000, generator=g))       # magnitudes over ~30 decades
    v = torch.cat([v, torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.1754944e-30, 65504.0, 1.0 + 2.0 ** -23])])
    p0, p1, p2, r2 = pieces(v)
    assert torch.equal(p2, r2), 'the third remainder has at most 8 significant bits: its bf16 rounding is exact'
    assert torch.equal((p0.double() + p1.double() + p2.double()).float(), v) and torch.equal(p0.double() + p1.double() + p2.double(), v.double())
    nz = v != 0
    assert float((p1[nz].abs() / v[nz].abs()).max()) <= 2.0 ** -8 and float((p2[nz].abs() / v[nz].abs()).max()) <= 2.0 ** -16
    assert torch.equal(pieces(torch.zeros(4))[1], torch.zeros(4))       # exact zeros stay exact zeros (the EBP gates rely on them)


def test_six_products_reproduce_the_fp32_product():
    g = torch.Generator().manual_seed(1)
    a = torch.randn(100000, generator=g)
    b = torch.randn(100000, generator=g).clamp_min(0) * 3
    pa, pb = pieces(a)[:3], pieces(b)[:3]
    exact = a.double() * b.double()
    six = sum(pa[i].double() * pb[j].double() for i in range(3) for j in range(3) if i + j <= 2)
    for i in range(3):
        for j in range(3):
            prod = pa[i].double() * pb[j].double()
            assert torch.equal(prod.float().double(), prod), 'a bf16 x bf16 product has 16 significant bits: exact in fp32'
    nz = exact != 0
    rel = ((six - exact).abs() / exact.abs())[nz]
    assert float(rel.max()) <= 2.0 ** -23 and float(rel.mean()) <= 2.0 ** -26
    three = sum(pa[i].double() * pb[j].double() for i in range(2) for j in range(2) if i + j <= 1)
    assert float(((three - exact).abs() / exact.abs())[nz].max()) > 2.0 ** -18      # bf16x3 is a different class: tests/precision/split_probe.py
