"""Dev tool: compile medtok_vq.hip to gfx950 asm and print the memory/MFMA skeleton of a kernel's hot loop.
    python tools/asm_loop.py <mangled-name-substring> [lines_before_first_mfma]"""
import re, subprocess, sys, collections
sub = sys.argv[1]; before = int(sys.argv[2]) if len(sys.argv) > 2 else 60
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S", "--cuda-device-only",
                       "-o", "/tmp/vq.s", "medtok_amd/csrc/medtok_vq.hip"], stderr=subprocess.DEVNULL)
s = open("/tmp/vq.s").read()
names = re.findall(r"^(_Z\w+):", s, re.M)
name = [n for n in names if sub in n][0]
body = s[s.index(name + ":"):]
body = body[:body.index(".end_amdhsa_kernel")]
lines = [l for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
idx = [i for i, l in enumerate(lines) if "v_mfma" in l]
res, run = [], 0
for l in lines[max(0, idx[0] - before): idx[0] + 3]:
    t = l.strip(); op = t.split()[0]
    if op.startswith(("v_mfma", "ds_", "global_", "buffer_", "s_waitcnt", "s_barrier", "s_cbranch", "s_branch")) or t.endswith(":"):
        if run: res.append(f"   [{run} other]"); run = 0
        res.append(t[:90])
    else:
        run += 1
print(name); print("\n".join(res))
for key in (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".group_segment_fixed_size"):
    m = re.search(re.escape(name) + r".*?" + re.escape(key) + r":\s+(\d+)", s, re.S)
print(re.findall(r"\.name:\s+" + re.escape(name) + r".*?\.vgpr_count:\s+(\d+)", s, re.S)[:1], "vgprs")
