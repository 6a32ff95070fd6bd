#!/bin/bash
# one summary line of a bench run: tools/lab/bench_line.sh [bench.py flags]
python bench.py "$@" 2>/dev/null | tail -1 > /tmp/bench_line.json && python tools/lab/show_bench.py /tmp/bench_line.json
