#!/bin/bash
# Proof that __graft_entry__.build() compiles everything from a CLEAN checkout (no prebuilt *.so / *.o travels in git):
# exports HEAD into a scratch directory, runs build() there and lists what it produced.
#     bash tools/clean_build_check.sh > profiles/r05_clean_build.txt
set -e -o pipefail
ROOT=$(git rev-parse --show-toplevel)
HEAD=$(git -C "$ROOT" rev-parse HEAD)
TMP=$(mktemp -d /tmp/ocd_clean_XXXXXX)
trap 'rm -rf "$TMP"' EXIT
git -C "$ROOT" archive "$HEAD" | tar -x -C "$TMP"
echo "# clean checkout of $HEAD in $TMP: $(find "$TMP" -name '*.so' -o -name '*.o' | wc -l) prebuilt binaries in the tree"
cd "$TMP"
START=$(date +%s)
python3 -c "import __graft_entry__ as g; g.build(); print('build() returned')"
END=$(date +%s)
echo "# build() took $((END - START)) s; it produced:"
find . -name '*.so' -printf '%s bytes  %p\n' | sort -k3
python3 - <<'PY'
import ctypes, sys
sys.path.insert(0, ".")
from l4dc_mpc_ocd_amd import abi
lib = abi.load_hip_library()
print("# libocd_hip.so of the clean build: ABI", lib.ocd_abi_version(), "-- every symbol of include/ocd.h bound;",
      "kernel sources", abi.kernel_source_sha())
PY
