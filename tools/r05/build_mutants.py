"""Dev tool (this container): variant builds of the library made by TEXT SUBSTITUTION on a copy of the sources (the product sources
carry no build-time switch), into devlib/<name>/libmedtok_vq.so (git-ignored; travels to the GPU box with gpurun).

    python tools/r05/build_mutants.py [name ...]

  nowait     the statement that holds the MFMA -> VALU wait states in front of the filter kernels' epilogues removed
             (round 4's 335-of-600 000-rows bug): tests/fuzzers.py must FAIL on it
  d32stage   rows of <= 32 elements padded to 32, not 64, columns and sent to the general filter kernel (one 32-deep stage per code
             tile: round 4's start-value race): tests/fuzzers.py must FAIL on it
  halfbar    TIMING ONLY (wrong results): filter_f16_kernel with its stage barrier and copy wait in every OTHER stage only
  k64emu     TIMING ONLY (wrong results): halfbar + the copies of two stages issued together in every other stage -- the instruction
             stream of a 64-deep stage (half the barriers, bursts of eight copies per wave) on the 32-deep ring
  nobar      TIMING ONLY (wrong results): no stage barrier at all
  rr16, rr8  the re-score kernel with 16 / 8 rows per block instead of 32 (same results)
  ms_pb1, ms_pb4   the batched small searches with about one / four code tiles per block instead of two (same results)
  rw_128     the few-rows re-score with 128 elements of the code row per chain step instead of 64 (same results)
  rw_nolds   the few-rows re-score with its chains reading the x row from global memory, 32 elements of both rows per step (same results)
  eps_emul   TIMING ESTIMATE (results right only because the actual errors are far inside): the window's operand-rounding term at 0.42 of
             its worst case -- what a bound from measured rounding residuals would typically give
  eps21      the shortlist window with rounds 1-4's accumulation budget (D 2^-21 instead of D 2^-22; same results, more candidates)
  gemm_mt4   the dense products always on 256- (or 192-) feature tiles, as before the 128- / 64-feature tiles for small products (same results)
  n_noscan   TIMING ONLY (wrong results): filter_rows64n_kernel without its scans (MFMAs, copies, barriers and start-value reads only)
  n_l<A>m<B> filter_rows64n_kernel learning from A code tiles (16 in the product) and recomputing the limits after every B-th scanned
             tile (1 in the product); same results
"""
import shutil, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from medtok_amd.csrc import build as B

def sub(text, old, new, count=1):
    assert text.count(old) >= 1, f"pattern not found: {old[:60]!r}"
    return text.replace(old, new, count)

def mutate(name, src):
    f = src / "filter_f16.h"
    t = f.read_text()
    hip = src / "medtok_vq.hip"
    h = hip.read_text()
    if name == "nowait":
        t = t.replace('asm volatile("s_nop 15\\n\\ts_nop 1"', 'asm volatile(""')
        assert 's_nop 15' not in t
    elif name == "d32stage":
        h = sub(h, "f.dp = (int)lmax(2 * F_BK, (d + F_BK - 1) / F_BK * F_BK);", "f.dp = (int)lmax(F_BK, (d + F_BK - 1) / F_BK * F_BK);")
    elif name in ("halfbar", "k64emu", "nobar"):
        old = '''        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        tick(1);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        tick(2);'''
        if name == "nobar":
            new = '''        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        tick(1);
        asm volatile("" ::: "memory");
        tick(2);'''
        else:
            new = '''        if (!(s & 1)) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        tick(1);
        if (!(s & 1)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        tick(2);'''
        t = sub(t, old, new)
        if name == "k64emu":
            t = sub(t, "        if (late && s > 0) stage();         // waves 4-7: stage s+2 (slot s-2, free since the barrier of iteration s-1)",
                    "        if (late && s > 0 && !(s & 1)) { stage(); stage(); }")
            t = sub(t, "        if (!late) stage();             // waves 0-3: stage s+3 (slot s-1: everyone is past reading it)",
                    "        if (!late && !(s & 1)) { stage(); stage(); }")
    elif name in ("rr16", "rr8"):
        # TIMING + results stay right: the re-score's rows per block (32 in the product)
        rr = name[2:]
        h = sub(h, "hipLaunchKernelGGL((rescore_kernel<T, 32>), dim3((unsigned)((n + 31) / 32)), dim3(256), 0, s, MEDTOK_RESCORE_ARGS);",
                f"hipLaunchKernelGGL((rescore_kernel<T, {rr}>), dim3((unsigned)((n + {rr} - 1) / {rr})), dim3(8 * {rr}), 0, s, MEDTOK_RESCORE_ARGS);")
    elif name in ("ms_pb1", "ms_pb4"):
        f_ = {"ms_pb1": "4L", "ms_pb4": "1L"}[name]
        h = sub(h, "const long per_block = lmax(1, (tiles_total + 2L * di.cus - 1) / (2L * di.cus));", f"const long per_block = lmax(1, (tiles_total + {f_} * di.cus - 1) / ({f_} * di.cus));")
    elif name == "rw_128":
        t = sub(t, "            for (; i + 64 <= d; i += 64) {\n                float4 wq[16];", "            for (; i + 128 <= d; i += 128) {\n                float4 wq[32];")
        t = sub(t, "                for (int j = 0; j < 16; ++j) wq[j] = ld4(wr + i + 4 * j);", "                for (int j = 0; j < 32; ++j) wq[j] = ld4(wr + i + 4 * j);")
        t = sub(t, "                for (int j = 0; j < 16; j += 2) {\n                    const float4 xa = *reinterpret_cast<const float4 *>(xs + i + 4 * j)", "                for (int j = 0; j < 32; j += 2) {\n                    const float4 xa = *reinterpret_cast<const float4 *>(xs + i + 4 * j)")
    elif name == "rw_nolds":
        t = sub(t, "    const bool x_lds = d <= RW_XMAX;", "    const bool x_lds = false;")
    elif name == "eps_emul":
        t = sub(t, "return 0x1p-10f + 0x1p-20f + (float)d * 0x1p-22f; }", "return 0.42f * 0x1p-10f + 0x1p-20f + (float)d * 0x1p-22f; }")
    elif name == "eps21":
        t = sub(t, "return 0x1p-10f + 0x1p-20f + (float)d * 0x1p-22f; }", "return 0x1p-10f + 0x1p-20f + (float)d * 0x1p-21f; }")
        t = sub(t, "const float start = (float)(d + 64) * 0x1p-22f * en_max;", "const float start = (float)(d + 64) * 0x1p-21f * en_max;")
    elif name == "gemm_mt4":
        h = sub(h, "    if (!mt3 && row_ids * ((n_g + 255) / 256) * groups < half_chip) mt = row_ids * ((n_g + 127) / 128) * groups < half_chip ? 1 : 2;\n", "    (void)half_chip;\n")
    elif name == "n_noscan":
        t = sub(t, "            filter_scan<TOPK, false, true>(row, acc[M], cb_, cbase, multi, nullptr, cc, bias);                            \\\n", "            (void)cb_;                                                                                               \\\n")
        t = sub(t, "            thr_insert_med3<TOPK>(row.tv, u == u ? u : INFINITY);\n            R64N_INIT_LDS(m, st + 1);", "            (void)u;\n            R64N_INIT_LDS(m, st + 1);")
    elif name.startswith("n_l") and "m" in name[3:]:
        A, B_ = name[3:].split("m")
        t = sub(t, "constexpr int R64N_LEARN = 16;", f"constexpr int R64N_LEARN = {int(A)};")
        if int(B_) > 1:
            t = sub(t, "        if (st < nct) filter_merge_halves<TOPK>(row, lh);\n        init_wait();",
                    f"        if (st < nct && (st - W < 4 || (st - W) % {int(B_)} == {int(B_)} - 1 || st == nct - 1)) filter_merge_halves<TOPK>(row, lh);\n        init_wait();")
    else:
        raise SystemExit(f"unknown mutant {name}")
    f.write_text(t); hip.write_text(h)

def main(names):
    for name in names:
        out = ROOT / "devlib" / name
        src = out / "src" / "medtok_amd" / "csrc"
        if out.exists(): shutil.rmtree(out)
        src.mkdir(parents=True)
        for p in B.HERE.glob("*.h"): shutil.copy(p, src / p.name)
        shutil.copy(B.SRC, src / B.SRC.name)
        (out / "src" / "include").mkdir()
        shutil.copy(B.HEADER, out / "src" / "include" / B.HEADER.name)
        mutate(name, src)
        so = out / "libmedtok_vq.so"
        subprocess.check_call([B.hipcc(), *B.FLAGS, str(src / B.SRC.name), "-o", str(so)])
        shutil.rmtree(out / "src")
        print("built", so)

if __name__ == "__main__":
    main(sys.argv[1:] or ["nowait", "d32stage", "halfbar", "k64emu", "nobar"])
