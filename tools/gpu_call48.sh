#!/bin/bash
timeout 150 python tools/step_ab.py --passes 5 --arm auto=0,0,0,0 --arm p4=10,10,10,10 --arm w4_256=13,13,13,13 2>&1 | grep -v amdgpu.ids | tail -3
