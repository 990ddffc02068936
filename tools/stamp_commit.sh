#!/bin/bash
# records the commit (and whether the tree is dirty) together with the hash of the shipped sources for runs on the GPU
# box, where .git does not travel; bench.py trusts the stamp only while that hash equals the sources it runs on
cd "$(dirname "$0")/.." && echo "$(git rev-parse --short HEAD)$(git diff --quiet HEAD -- . ':!PROGRESS.jsonl' || echo +dirty) $(python3 bench.py --source-hash)" > .bench_commit && cat .bench_commit
