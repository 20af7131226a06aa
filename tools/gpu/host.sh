#!/bin/bash
# host share of a rank batch on the GPU box's cores: resolve_requests with 1 / 8 / 32 threads (tools/host_bench.py)
cd "$GRAFT_REPO_ROOT"
for t in 1 8 32; do python tools/host_bench.py c2 3840 $t 2>&1 | tail -2 | head -1; done
