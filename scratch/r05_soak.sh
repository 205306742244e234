#!/bin/bash
# r05 soak of the eval forward: inputs rewritten in place before every call, outputs compared bitwise with the first pass -- the rule's
# own layout and forced layouts, gates and events, bf16 and fp32 features (k_img_pool / k_img_pool32), the shipped configuration
run() { echo "[$1] $(env $1 SOAK_B=$4 python scratch/soak_fork.py $2 $3 $5 2>&1 | tail -1)"; }
run "PTX_GATE=1" cfg2 1500 4
run "PTX_GATE=0" cfg2 800 4
run "PTX_GATE=1" cfg2 800 4 f32
run "PTX_LAYOUT=1---" cfg2 800 2
run "PTX_LAYOUT=0--1" cfg2 800 4
run "PTX_LAYOUT=0--0" cfg2 800 6
run "PTX_GATE=1" cfg2 400 32
run "PTX_GATE=1" cfg4 800 6 f32
run "PTX_LAYOUT=11--" cfg4 600 3 f32
run "PTX_LAYOUT=100-" cfg4 600 6 f32
run "PTX_GATE=0" cfg4_room 600 6 f32
run "PTX_GATE=1" cfg4_room 800 6 f32
run "PTX_GATE=1" cfg1 800 1 f32
run "PTX_GATE=1" cfg5 300 2
