set -u
export TMPDIR=/tmp ADYPT_CACHE=/tmp/adypt_cache
mkdir -p gpurun_out/r5
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r5/gputest.log 2>&1; rc=$?; echo "pytest rc $rc" >> gpurun_out/r5/gputest.log; tail -3 gpurun_out/r5/gputest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
