#!/bin/bash
python bench.py --no-cpu-baseline "$@" 2>/dev/null < /dev/null | python scratch/benchline.py
