#!/bin/bash
# local helper: build the library here (hipcc cross-compiles without a GPU; a stale .so is refused on the GPU box), then hand the command to gpurun
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" | tail -1
exec /usr/local/graft/bin/gpurun "$@"
