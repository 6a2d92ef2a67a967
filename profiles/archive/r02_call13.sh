cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for d in 3 3 3 1 2 4; do LP_MAX_STEPS_IN_FLIGHT=$d timeout 300 python bench.py --no-cpu-baseline --no-profile --no-secondary --steps 12 > gpurun_out/r02m_bench_d$d.json.log 2>&1; echo -n "depth $d: "; tail -1 gpurun_out/r02m_bench_d$d.json.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['config']['memory'])"; done
