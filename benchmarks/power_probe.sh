#!/bin/bash
# Package power and shader clock while a command runs (rocm-smi sampled every 0.7 s from the 6th second on):
#   power_probe.sh LABEL command...        (on a GPU box; evidence for DESIGN 8-0: the fp16x3 encoder forward at 256 x 512
#   tokens sits at the 1400 W package cap with the shader clock pulled down to ~2.0 GHz)
label=$1; shift
"$@" > /dev/null 2>&1 &
PID=$!
sleep 6
echo "== $label: $*"
for i in 1 2 3 4 5 6; do
  kill -0 $PID 2>/dev/null || break
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "Package Power\|sclk" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 0.7
done
wait $PID
