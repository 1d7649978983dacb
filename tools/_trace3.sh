cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o p -- python3 bench.py --no-kernel-timing --no-cpu-baseline --no-side-workloads --sample-iters 1 > /tmp/tr.log 2>&1 || { tail -5 /tmp/tr.log; exit 1; }
python3 tools/step_trace.py /tmp/tr 16 17 3 16 1 > gpurun_out/s5_trace3.txt 2>&1; python3 tools/step_trace.py /tmp/tr 10 20 > gpurun_out/s5_trace3_steps.txt 2>&1; python3 tools/step_trace.py /tmp/tr 16 17 3 16 1 250 0 1100 > gpurun_out/s5_trace3_range.txt 2>&1
grep -c . gpurun_out/s5_trace3.txt
