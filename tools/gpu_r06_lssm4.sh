#!/bin/bash
timeout 900 python tools/lssm_d8_ab.py 2>&1 | grep -v amdgpu.ids
