#!/bin/bash
# Samples socket power / sclk with rocm-smi while a command runs:  tools/power_probe.sh <label> <cmd...>
label=$1; shift
"$@" > /dev/null 2>&1 &
pid=$!
sleep 2.5
n=0
while kill -0 $pid 2>/dev/null && [ $n -lt 12 ]; do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' ' | sed "s/^/$label: /"; echo
  n=$((n+1)); sleep 0.7
done
wait $pid
