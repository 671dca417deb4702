#!/bin/bash
timeout 150 python tools/step_ab.py --passes 5 --arm auto=0,0,0,0 --arm gu_p4=0,0,10,0 --arm qkv_p4=10,0,0,0 2>&1 | grep -v amdgpu.ids | tail -2
