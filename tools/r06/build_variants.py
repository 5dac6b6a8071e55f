"""Dev tool (this container): round-6 variant builds of the library by TEXT SUBSTITUTION on a copy of the sources, into
devlib/<name>/libmedtok_vq.so (git-ignored; travels to the GPU box with gpurun).  Same mechanism as tools/r05/build_mutants.py.

    python tools/r06/build_variants.py [name ...]

  wv_epi   TIMING ONLY (wrong results): shared_kv_attention_pp_kernel with what a fused W_v,h epilogue would add at the least -- per
           32-row tile a (hi, lo) slice of W_v,h (192 x 768 x 4 B = 590 KB = 12 ring chunks of 48 KB) streamed L2 -> LDS through the
           kernel's own key ring (copy, wait, block-wide barrier per chunk) and the product's matrix work (32 x 192 x 768, three fp16
           passes: 18 v_mfma_f32_32x32x16_f16 per wave and chunk).  Not emulated (so the estimate is a LOWER bound of the cost): the
           context tile's way from accumulators to MFMA A operands through LDS, and the cross-wave reduction of the partial sums.
           Paired with tools/r06/ab_wv_epilogue.py, which also drops the W_v GEMM launch the fusion would save.
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
    if name == "wv_epi":
        f = src / "attention_pp.h"
        t = f.read_text()
        old = '''#undef PP_MFMA16
#undef PP_MFMA32
    if (TIMED && dbg && lane == 0) {'''
        new = '''    // ---- EMULATION (tools/r06/build_variants.py wv_epi): the W_v,h slice through the ring + the product's MFMAs
    if (active || true) {
        const half8 ea = qh[0][0], eb = qlo[0][0];
        for (int e = 0; e < 12; ++e) {
            if (nchunk > 0) stage(e % nchunk);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
                acc[tt] = PP_MFMA32(ea, eb, acc[tt], 0, 0, 0);
                acc[tt] = PP_MFMA32(eb, ea, acc[tt], 0, 0, 0);
                acc[tt] = PP_MFMA32(ea, ea, acc[tt], 0, 0, 0);
            }
            __builtin_amdgcn_s_barrier();
        }
    }
#undef PP_MFMA16
#undef PP_MFMA32
    if (TIMED && dbg && lane == 0) {'''
        t = sub(t, old, new)
        f.write_text(t)
    else:
        raise SystemExit(f"unknown variant {name}")


def main(names):
    for name in names:
        out = ROOT / "devlib" / name
        src = out / "src" / "medtok_amd" / "csrc"
        if out.exists():
            shutil.rmtree(out)
        src.mkdir(parents=True)
        for p in B.HERE.glob("*.h"):
            shutil.copy(p, src / p.name)
        shutil.copy(B.SRC, src / B.SRC.name)
        (out / "src" / "include").mkdir()
        shutil.copy(B.HEADER, out / "src" / "include" / B.HEADER.name)
        mutate(name, src)
        so = out / "libmedtok_vq.so"
        subprocess.check_call([B.hipcc(), *B.FLAGS, str(src / B.SRC.name), "-o", str(so)])
        shutil.rmtree(out / "src")
        print("built", so)


if __name__ == "__main__":
    main(sys.argv[1:] or ["wv_epi"])
