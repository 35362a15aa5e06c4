set -u
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/c14_tests.log 2>&1; echo "tests rc $?"; tail -3 gpurun_out/c14_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
