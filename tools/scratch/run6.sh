#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
hipcc -O3 --offload-arch=gfx950 -Wno-unused-result tools/bench_src/pk_fma_chain.hip -o /tmp/pk_chain 2>/dev/null && /tmp/pk_chain
