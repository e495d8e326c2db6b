"""The three LDS tenants of a k_describe wave (raw window, row pairs of the horizontal pass, blurred patch) overlap in ONE region
(orbx_kernels.hip: DW_RAW0).  This test re-derives, from the constants in the source, that no phase overwrites data a later phase
(or a later trip / iteration of the same phase) still reads -- the argument of the kernel's comment, executed.  CPU only."""
import os
import re

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "orb_slam3_fast_amd", "csrc", "orbx_kernels.hip")


def constants():
    text = open(SRC).read()
    env = {}
    for name in ("DW_ROWS", "DW_RP", "DW_HP", "DW_BP", "DW_RAW0", "DW_WAVE_DW"):
        m = re.search(r"constexpr int %s = ([^;]+);" % name, text)
        assert m, name
        env[name] = int(eval(m.group(1), {}, env))
    return env


def test_window_fits_the_region():
    c = constants()
    # the window plus the slack row the horizontal pass reads (second row of pair 21) lies inside the wave's region
    assert c["DW_RAW0"] + c["DW_ROWS"] * c["DW_RP"] + 16 <= 23 * c["DW_HP"] <= c["DW_WAVE_DW"]
    # row pair 22 (read by the vertical pass for padding rows only) is addressable
    assert 23 * c["DW_HP"] <= c["DW_WAVE_DW"]
    # a workgroup of four waves takes 12 of the CU's 128 LDS granules of 1280 bytes: eight workgroups per CU
    granules = -(-4 * c["DW_WAVE_DW"] * 4 // 1280)
    assert granules == 12 and 128 // granules >= 8


def test_horizontal_pass_never_overwrites_rows_still_to_be_read():
    c = constants()
    hp, rp, raw0 = c["DW_HP"], c["DW_RP"], c["DW_RAW0"]
    for t in range(4):                      # trip t: lanes (j0 = 0..5, q = 0..9) handle row pair 6 t + j0
        pairs = [6 * t + j0 for j0 in range(6) if not (t == 3 and j0 >= 4)]
        write_end = max(hp * (p + 1) for p in pairs)                      # row pairs written in this trip end here (dwords)
        later_rows = [r for tt in range(t + 1, 4) for j0 in range(6) if not (tt == 3 and j0 >= 4) for r in (2 * (6 * tt + j0), 2 * (6 * tt + j0) + 1)]
        if later_rows:
            read_begin = raw0 + rp * min(later_rows)                      # first dword a later trip reads
            assert write_end <= read_begin, (t, write_end, read_begin)
    # (inside a trip every lane reads its rows before any lane writes its row pair: one wave, program order)


def test_vertical_pass_never_overwrites_row_pairs_still_to_be_read():
    c = constants()
    hp, bp = c["DW_HP"], c["DW_BP"]
    items = list(range(5 * 37))
    iters = [items[k:k + 64] for k in range(0, len(items), 64)]           # for (i = lane; i < 185; i += 64)
    for k, cur in enumerate(iters):
        written_end = max((8 * (i // 37) + 8) * bp for it in iters[:k + 1] for i in it) // 4   # patch rows 8 ch .. 8 ch + 7, dwords
        later = [i for it in iters[k + 1:] for i in it]
        if later:
            read_begin = min(hp * (4 * (i // 37)) for i in later)         # row pairs 4 ch .. 4 ch + 6
            assert written_end <= read_begin, (k, written_end, read_begin)
