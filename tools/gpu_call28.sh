#!/bin/bash
for v in 0 4 8 0 4; do echo "== ABL $v"; VT_FLASH_ABL=$v timeout 120 python tools/attn_bench.py 2>&1 | grep '"S"' | head -2; done
