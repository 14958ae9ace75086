#!/bin/bash
# any-length n_fft sweep (mixed-radix / Bluestein inside the fused kernel) against the neighbouring powers of two
# usage (on the GPU box): bash tools/sweep_anylen.sh > gpurun_out/anylen_sweep.txt
set -e
for cfg in "1024 256" "1000 250" "512 128" "500 125" "2048 512" "1536 384" "1920 480" "768 192" "1001 250" "999 250" "4096 1024" "3000 750"; do
  set -- $cfg
  python tools/bench_stft.py --dtype f32 --n-fft $1 --hop $2 --batch 64 --steps 10
done
python tools/bench_stft.py --dtype f64 --n-fft 1024 --hop 256 --batch 64 --steps 5
python tools/bench_stft.py --dtype f64 --n-fft 1000 --hop 250 --batch 64 --steps 5
