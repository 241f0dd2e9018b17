#!/bin/bash
# Samples sclk / socket power with rocm-smi while a command runs:  bash tools/clock_sample.sh <out.txt> <command...>
out=$1; shift
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket Graphics Package Power|Average Graphics Package Power|Current Socket" | tr '\n' ' '; echo; sleep 0.25; done ) > $out &
SAMPLER=$!
"$@"
kill $SAMPLER
