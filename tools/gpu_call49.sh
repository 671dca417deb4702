#!/bin/bash
timeout 150 python tools/step_ab.py --passes 5 --arm auto=0,0,0,0 --arm qkv256=13,0,0,0 --arm gu320=0,0,14,0 --arm o_down_split=0,13,0,13 2>&1 | grep -v amdgpu.ids | tail -2
