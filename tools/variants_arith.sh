#!/bin/bash
# Diagnostic: tools/variants.py with the HBM stream removed (diagnostics library, MOLANN_DEBUG_ABLATE=64): the
# arithmetic cost of each stage of the lane kernel at full occupancy
cd "$(dirname "$0")/.."
MOLANN_DIAG_LIB=1 MOLANN_DEBUG_ABLATE=${ABLATE:-64} python tools/variants.py 2>/dev/null
