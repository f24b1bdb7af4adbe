# round 3, call ZZQ: tiling 6 forced onto the q/k/v projection and the 64x64-level FF1 (table copy), per-shape times
mkdir -p gpurun_out/r3zzq
for tb in tweediemix_amd/tuned_gfx950.json tools/ab/t6_table.json; do
  TMIX_TUNE_FILE=$tb TMIX_BENCH_SHAPES=1 timeout 600 python bench.py --kind lora --no-cpu-baseline --no-video --no-trajectory --steps 40 2>gpurun_out/r3zzq/s.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],3), d['parity_check']['rel_l2'])"
  grep -E "'gemm', 4, 1024, 3840|'gemm', 1, 16384, 5120" gpurun_out/r3zzq/s.err | cut -c1-120
done
