"""Static instruction counts per kernel section (between SSQ_STAMP markers) of a -DSSQ_MARK build:
    hipcc -std=c++17 -O3 --offload-arch=gfx950 -DSSQ_MARK -S --cuda-device-only -o /tmp/m.s ssqueeze_rs_amd/csrc/stft_fused.hip
    python tools/count_sections.py /tmp/m.s _ZN3ssq17stft_fused_kernelIfLi10ELb1EEEvNS_7StftDevIT_EE
Counts the straight-line interior path: the text between marker i and the next marker."""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith(name + ":"))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end]
marks = [(i, int(re.search(r"SSQ_SECTION (\d+)", l).group(1))) for i, l in enumerate(body) if "SSQ_SECTION" in l]
names = {0: "top..window mult", 1: "decode+load issue", 2: "FFT", 3: "shuffles", 4: "unpack+phase+bins",
         5: "reduce+scale", 6: "atomics", 7: "barrier1", 8: "readout", 9: "barrier2"}
print("markers at", marks)
for (a, sa), (b, sb) in zip(marks, marks[1:] + [(len(body), -1)]):
    cnt = collections.Counter()
    for l in body[a:b]:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m:
            continue
        op = m.group(1)
        if op.startswith("v_pk_"):
            cnt["valu_pk"] += 1
        elif op.startswith("v_"):
            cnt["valu"] += 1
        elif op.startswith("s_waitcnt"):
            cnt["waitcnt"] += 1
        elif op.startswith("s_"):
            cnt["salu"] += 1
        elif op.startswith("ds_"):
            cnt["lds"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            cnt["vmem"] += 1
    tot = sum(cnt.values())
    print(f"after marker {sa:2d} (-> {names.get(sb, '?'):22s}): total {tot:5d}  " + "  ".join(f"{k}={v}" for k, v in sorted(cnt.items())))
