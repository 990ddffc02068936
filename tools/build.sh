#!/bin/bash
# builds every native part in-tree (same as __graft_entry__.build()); exit code != 0 on failure
cd "$(dirname "$0")/.." && python -c "import __graft_entry__ as g; g.build()" "$@"
