#!/bin/bash
# Developer tool (GPU box): runs the steps given on stdin one after the other ("<seconds> <log name> <command...>"
# per line) under `timeout -k 10`, each with its output under gpurun_out/$1/; a step that is KILLED at its limit
# ends the call (no further GPU step after a hang), a step that merely fails does not.
# usage: tools/gpu_steps.sh <out dir under gpurun_out> < steps.txt
D=gpurun_out/$1
mkdir -p $D
while read -r limit name cmd; do
  [ -z "$limit" ] && continue
  case "$limit" in \#*) continue ;; esac
  echo "== $name: $cmd"
  timeout -k 10 $limit bash -c "$cmd" > $D/$name.log 2>&1 < /dev/null
  rc=$?
  echo "== $name rc $rc: $(tail -1 $D/$name.log | cut -c1-300)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "== $name hit its limit: stopping"; exit 1; fi
done
exit 0
