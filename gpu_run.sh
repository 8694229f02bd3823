#!/bin/sh
# developer helper: rebuild every native piece, then run a command on the MI355X box
set -e
cd "$(dirname "$0")"
python -c "import __graft_entry__ as g; g.build()"
exec /usr/local/graft/bin/gpurun --timeout "${GPU_TIMEOUT:-1200}" -- "$@"
