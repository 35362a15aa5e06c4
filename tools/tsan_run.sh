#!/bin/bash
# The multi-threaded callers (worker pools, one fnx ctx per thread) against the ThreadSanitizer build of the host layer
# (make -C fennec_amd/csrc tsan).  Reports are kept only when a frame of libfennec_hip is in them: python, torch and the
# HIP runtime are not instrumented and TSan cannot see their synchronisation.
#   gpurun -- 'bash tools/tsan_run.sh'
set -u
cd "$(dirname "$0")/.."
RT=$(gcc -print-file-name=libtsan.so)
[ -f fennec_amd/libfennec_hip_tsan.so ] || { echo "build it first: make -C fennec_amd/csrc tsan"; exit 2; }
cp fennec_amd/libfennec_hip.so /tmp/libfennec_hip.keep
cp fennec_amd/libfennec_hip_tsan.so fennec_amd/libfennec_hip.so
mkdir -p gpurun_out
export LD_LIBRARY_PATH=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))"):${LD_LIBRARY_PATH:-}
export TSAN_OPTIONS=halt_on_error=0:report_signal_unsafe=0:history_size=4:log_path=gpurun_out/tsan_report:exitcode=0
# gcc 11's TSan does not map under high-entropy ASLR: run with it off
timeout 1500 setarch "$(uname -m)" -R env LD_PRELOAD="$RT $(gcc -print-file-name=libstdc++.so)" python tools/tsan_workload.py > gpurun_out/tsan_workload.log 2>&1
rc=$?
cp /tmp/libfennec_hip.keep fennec_amd/libfennec_hip.so
tail -5 gpurun_out/tsan_workload.log; echo "workload rc=$rc"
python - <<'PY'
import glob, re
reps = []
for f in glob.glob("gpurun_out/tsan_report*"):
    reps += open(f, errors="replace").read().split("WARNING: ThreadSanitizer")[1:]
ours = 0
for r in reps:
    for sec in re.split(r"\n\s*\n", r):
        if re.search(r"^\s*(Read|Write|Previous|Atomic)", sec, re.M):
            # the accessing code: the first frame that is not the sanitizer's own interceptor
            frames = [m for m in re.findall(r"#\d+ (.*)", sec) if "libtsan" not in m]
            if frames and "libfennec_hip" in frames[0]:
                ours += 1
                print(r[:1500])
                break
print(f"ThreadSanitizer reports: {len(reps)}; with the racing access in libfennec_hip: {ours} "
      "(the rest: allocator / mutex traffic inside the uninstrumented HIP runtime and python)")
PY
exit $rc
