#!/bin/bash
# round 6 on the GPU box: STAGES="tests bench ..." (each writes under gpurun_out/r06/)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; out=gpurun_out/r06; mkdir -p $out
for st in ${STAGES:-tests bench}; do
  case $st in
    tests) ( time timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -x -q -m gpu ${PYTEST_ARGS:-} ) > $out/pytest_gpu_${TAG:-run}.log 2>&1; tail -5 $out/pytest_gpu_${TAG:-run}.log ;;
    bench) ( time timeout 1700 python bench.py --gpus 1 --steps 20 --warmup 5 ${BENCH_ARGS:-} ) > $out/bench_${TAG:-run}.log 2>&1; grep '^{' $out/bench_${TAG:-run}.log > $out/bench_line_${TAG:-run}.json; tail -c 600 $out/bench_${TAG:-run}.log ;;
    quick) ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample-mbp 0 --e2e-mbp 0 --abi-mbp 0 --other-configs 0 --e2e-configs 0 ${BENCH_ARGS:-} ) > $out/quick_${TAG:-run}.log 2>&1; grep '^{' $out/quick_${TAG:-run}.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['roofline']['kernel_ms'])" ;;
    cfgs) for cfg in ${CFGS:-wgs30x wgs30x_mixed novaseq}; do A="--mode weak --config $cfg"; [ $cfg = tumor200x ] && A="--mode strong --contig-mbp 6.25"
            ( timeout 900 python bench.py --gpus 1 --steps 30 --warmup 5 --cpu-ref-mbp 0 --e2e-mbp 0 --abi-mbp 0 --other-configs 0 --e2e-configs 0 ${CFG_ARGS:-} $A ) > $out/cfg_${cfg}_${TAG:-run}.log 2>&1
            grep '^{' $out/cfg_${cfg}_${TAG:-run}.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$cfg', d['ms_per_step'], d['roofline']['kernel_ms'], (d.get('validated') or {}).get('full_contig'))"; done ;;
    e2e) for leg in ${LEGS:-sites tumor}; do ( time timeout 1500 python tools/e2e_configs.py --leg $leg ${E2E_ARGS:-} ) > $out/e2e_${leg}_${TAG:-run}.log 2>&1; tail -c 3000 $out/e2e_${leg}_${TAG:-run}.log; done ;;
    *) echo "unknown stage $st" ;;
  esac
done
