#!/bin/bash
# quick GPU check of the L2 matcher: parity tests, then the bench's isolated kernel time
timeout -k 10 300 python3 -u -m pytest tests/test_gpu_matching.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -3 || exit 1
timeout -k 10 200 python3 -u bench.py --steps 10 --cpu-pairs 0 --sustained-s 1 --streams ${1:-3} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('pairs/s', round(d['value']), 'sustained', round(d['sustained']['value']), 'filter_us', round(r['dominant_kernel_us'],1), 'contended', round(r['contended_kernel_us'],1), 'call_us', round(r['matcher_call_us'],1), 'frac', round(r['frac'],4))"
