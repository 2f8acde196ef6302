#!/bin/bash
# Diagnostic: phase shares and the shader clock the chip holds in the lane kernel, with and without its HBM stream
cd "$(dirname "$0")/.."
for ab in 0 64 256; do echo "== EXTRA_ABLATE=$ab (0: production data path, 64: no HBM stream after the first tile, 256: DMA from an L2-resident tile)"; EXTRA_ABLATE=$ab timeout -k 10 120 python tools/stamps.py C3 2>/dev/null; done
