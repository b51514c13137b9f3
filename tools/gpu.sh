#!/bin/bash
# Rebuild the HIP library, then run a command on the MI355X box through gpurun (a stale .so once measured the wrong code path):
#   bash tools/gpu.sh <timeout_s> '<command>'
set -e
make -C "$(dirname "$0")/../visualcloze_amd/csrc" -j8 2>&1 | grep -E "error|Error" && exit 1
python - <<'PY'
import os, sys
here = os.path.dirname(os.path.abspath("tools/gpu.sh"))
lib = "visualcloze_amd/lib/libvcloze_hip.so"
newest = max(os.path.getmtime(os.path.join("visualcloze_amd/csrc", f)) for f in os.listdir("visualcloze_amd/csrc") if f.endswith((".hip", ".h")))
newest = max(newest, os.path.getmtime("include/vcloze_hip.h"))
assert os.path.getmtime(lib) >= newest, "libvcloze_hip.so is older than its sources"
PY
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
