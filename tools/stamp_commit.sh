#!/bin/bash
# records the commit (and whether the tree is dirty) for runs on the GPU box, where .git does not travel
cd "$(dirname "$0")/.." && echo "$(git rev-parse --short HEAD)$(git diff --quiet HEAD -- . ':!PROGRESS.jsonl' || echo +dirty)" > .bench_commit && cat .bench_commit
