#!/bin/bash
# development: correctness on three scenes + 4K timing of the default pipeline (GPU box)
python -u tools/gpu_check.py scene4-specular scene5 scene7 2>&1 | grep -E "frame " | cut -c1-140
timeout -k 5 120 python -u tools/gpu_perf.py scene7 3840 2160 ${1:-6} noref noserial 2>/dev/null | grep -E 'frame :' | cut -c1-330
