#!/bin/bash
# Rebuild the HIP library, then run a command on the MI355X box through gpurun (a stale .so once measured the wrong code path):
#   bash tools/gpu.sh <timeout_s> '<command>'
set -e -o pipefail
cd "$(dirname "$0")/.."
make -C visualcloze_amd/csrc -j8 > /tmp/vc_make.log 2>&1 || { tail -n 30 /tmp/vc_make.log; echo "build failed" >&2; exit 1; }
python - <<'PY'
import os
lib = "visualcloze_amd/lib/libvcloze_hip.so"
newest = max(os.path.getmtime(os.path.join("visualcloze_amd/csrc", f)) for f in os.listdir("visualcloze_amd/csrc") if f.endswith((".hip", ".h")))
newest = max(newest, os.path.getmtime("include/vcloze_hip.h"))
assert os.path.getmtime(lib) >= newest, "libvcloze_hip.so is older than its sources"
PY
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
