#!/bin/bash
# dev (GPU box): run a command in the background and, when its log stops growing for 40 s, say from OUTSIDE where its
# threads are (state, wchan, kernel stack if readable), whether the GPU is busy, then end it (exact PID).
# usage: observe.sh <log> <command...>
LOG=$1; shift
"$@" > "$LOG" 2>&1 &
PID=$!
last=-1; still=0
for i in $(seq 1 60); do
    sleep 5
    kill -0 $PID 2>/dev/null || { wait $PID; echo "[observe] ended by itself, rc=$?"; exit 0; }
    sz=$(stat -c %s "$LOG")
    if [ "$sz" = "$last" ]; then still=$((still + 1)); else still=0; last=$sz; fi
    echo "[observe] t=$((i * 5)) s, log $sz bytes, quiet for $((still * 5)) s"
    if [ $still -ge 8 ]; then
        echo "[observe] stuck: process state"
        grep -E "State|Threads|SigQ|SigPnd|ShdPnd|SigBlk|SigIgn|SigCgt" /proc/$PID/status
        for t in /proc/$PID/task/*; do
            echo "  thread $(basename $t): $(cut -d' ' -f3 $t/stat) wchan=$(cat $t/wchan 2>/dev/null) comm=$(cat $t/comm)"
            cat $t/stack 2>/dev/null | head -8
        done
        rocm-smi --showuse 2>&1 | grep -i "busy\|use" | head -4
        top -b -n 1 -H -p $PID 2>/dev/null | tail -n +7 | head -12
        kill -TERM $PID; sleep 2; kill -CONT $PID 2>/dev/null; sleep 2; kill -KILL $PID 2>/dev/null
        wait $PID; echo "[observe] ended, rc=$?"
        exit 1
    fi
done
kill -KILL $PID; echo "[observe] time is up"
