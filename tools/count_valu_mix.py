"""Static VALU mix per kernel section (between SSQ_STAMP markers) of a -DSSQ_MARK build: the full-rate fp32 instructions two
waves of a SIMD can share (v_add/sub/mul/fma/fmac_f32, v_mov_b32: 2.3 cycles per wave-instruction with >= 2 waves,
profiles/r01_ubench_valu_rate*.txt) against everything else (>= 4.3: conversions, integer / 64-bit / bit ops, DPP, permlane,
readlane, v_rcp 8.3), plus the LDS / global instructions.
    hipcc -std=c++17 -O3 -fno-slp-vectorize --offload-arch=gfx950 -DSSQ_MARK -S --cuda-device-only -o /tmp/m.s \
          ssqueeze_rs_amd/csrc/stft_fused.hip -Iinclude
    python tools/count_valu_mix.py /tmp/m.s _ZN3ssq18stft_tx1024_kernelILb0ELb0ELi16ELb0EEEvNS_7StftDevIfEE
Sections: 9->0 window, 1->2 FFT (3 passes, exchanges, sample prefetch), 2->3 partner shuffles, 3->4 unpack + phase + bins,
4->5 column reduce + scale, 5->6 fixed-point scatter, 7->end read-out (BOTH read-out variants are in the text: the paired one
that runs is about half of it)."""
import collections, re, sys
src = open(sys.argv[1]).read().split("\n")
name = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith(name + ":"))
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end]
marks = [(i, int(re.search(r"SSQ_SECTION (\d+)", l).group(1))) for i, l in enumerate(body) if "SSQ_SECTION" in l]
FAST = ("v_add_f32","v_sub_f32","v_subrev_f32","v_mul_f32","v_fma_f32","v_fmac_f32","v_fmamk_f32","v_fmaak_f32","v_mov_b32")
for (a, sa), (b, sb) in zip(marks, marks[1:] + [(len(body), -1)]):
    fast = 0; slow = collections.Counter(); other = collections.Counter()
    for l in body[a:b]:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if not m: continue
        op = m.group(1)
        if op.startswith("v_"):
            base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
            if base in FAST and "dpp" not in op: fast += 1
            else: slow[base] += 1
        elif op.startswith(("ds_","global_","scratch_")): other[op] += 1
    print(f"sec {sa:2d}->{sb:2d}: fast={fast:4d} slow={sum(slow.values()):4d} {dict(slow)} | {dict(other)}")
