#!/bin/bash
# Round 3: the records the round-2 verdict asked for: counter passes of the hybrid sweep (configs[3]) and of the emission sweep
# (configs[4] on one GPU), of the one-launch forms of the brick sweep (dataflow 1 and 3), and a bench line on a tau ~ 1 field.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_records
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python tools/pmc_passes.py --out profiles/r03_pmc_dataflow1.json --kernels "brick_" --note "one launch, a workgroup per brick, global ticket, L2 write-back per brick" -- bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 --dataflow 1 > $OUT/df1.txt 2>&1 || { tail $OUT/df1.txt; exit 1; }
python tools/pmc_passes.py --out profiles/r03_pmc_dataflow2.json --kernels "brick_" --note "the same with write-through stores" -- bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 --dataflow 2 > $OUT/df2.txt 2>&1 || { tail $OUT/df2.txt; exit 1; }
python tools/pmc_passes.py --out profiles/r03_pmc_dataflow3.json --kernels "brick_" --note "persistent workgroups, a queue per XCD, queue_mix 2" -- bench.py --steps 1 --warmup 0 --no-cpu-baseline --lanes 1 --dataflow 3 --opt queue_mix=2 > $OUT/df3.txt 2>&1 || { tail $OUT/df3.txt; exit 1; }
cat $OUT/df1.txt $OUT/df2.txt $OUT/df3.txt
python tools/pmc_passes.py --out profiles/r03_pmc_config4.json --note "configs[3]: 128^3 + refined 32^3 block, 8 groups, 96 directions, 4 iterations (the first builds the plan) + one star" -- tools/bench_config4.py 128 > $OUT/c4.txt 2>&1 || { tail $OUT/c4.txt; exit 1; }
cat $OUT/c4.txt
python tools/pmc_passes.py --out profiles/r03_pmc_config5.json --note "configs[4] on one GPU: 256^3 x 8 x 96 source iterations, 3 of them" -- tools/bench_config5.py 256 3 > $OUT/c5.txt 2>&1 || { tail $OUT/c5.txt; exit 1; }
cat $OUT/c5.txt
timeout -k 10 300 python tools/bench_config4.py 128 > $OUT/config4_unprofiled.log 2>&1; grep "diffuse iteration\|tracer\|star" $OUT/config4_unprofiled.log | tail -8
timeout -k 10 300 python tools/bench_config5.py 256 6 > $OUT/config5_unprofiled.log 2>&1; grep "iteration" $OUT/config5_unprofiled.log | tail -4
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --tau-median 1.0 > $OUT/bench_tau1.json 2> $OUT/bench_tau1.err; python -c "
import json; d=json.load(open('$OUT/bench_tau1.json')); print('tau_median 1.0: step %.2f ms, sweep phase %.2f ms, value %.3e' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['value']))"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --in-process 2 --same-device > $OUT/bench_in_process2.json 2> $OUT/bench_in_process2.err; cut -c1-700 $OUT/bench_in_process2.json
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --in-process 4 --same-device --nnu 2 > $OUT/bench_in_process4.json 2> $OUT/bench_in_process4.err; cut -c1-700 $OUT/bench_in_process4.json
