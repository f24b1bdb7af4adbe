# round 3, call ZZP: tiling 6 (256 x 256 over FOUR waves of 128 x 128, accumulators in AGPRs) back as an epilogue-family instantiation: kernel tests, then a
# graph-timed refine that offers it to the 30 heaviest GEMM shapes, and the step with the table it produces
mkdir -p gpurun_out/r3zzp
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm" 2>&1 | tail -3
timeout 1500 python tools/refine_table.py tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzp/refined.json --top 30 --reps 9 --kinds lora --cands 6 > gpurun_out/r3zzp/refine.log 2>&1; echo "refine rc=$?"
grep -E "refine \(" gpurun_out/r3zzp/refine.log | awk -F': ' '{split($2,a," "); if (a[1]!=a[3]) print}' | cut -c1-170; grep refined gpurun_out/r3zzp/refine.log
for i in 1 2 3; do
for tb in tweediemix_amd/tuned_gfx950.json gpurun_out/r3zzp/refined.json; do
  TMIX_TUNE_FILE=$tb TMIX_BENCH_SHAPES=1 timeout 600 python bench.py --kind lora --no-cpu-baseline --no-video --no-trajectory --steps 40 2>gpurun_out/r3zzp/shapes_$(basename $tb .json).err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$tb', round(d['value'],2), round(d['ms_per_step'],3), {k: round(v['sum_launch_ms'],2) for k,v in d['roofline']['classes'].items()}, d['parity_check']['rel_l2'])"
done; done
for f in tuned_gfx950 refined; do echo $f; grep -E "'gemm', 4, 1024, 3840|'gemm', 1, 16384, 5120|'gemm', 4, 4096, 1920" gpurun_out/r3zzp/shapes_$f.err | cut -c1-120; done
