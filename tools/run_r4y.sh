#!/bin/bash
mkdir -p gpurun_out/r4y
(KRK_LSTM_V=3 timeout 200 python tools/ws_flake.py 1500 --slots 2>&1 | grep -v amdgpu.ids > gpurun_out/r4y/exchange_timeouts_three_in_flight.txt); cat gpurun_out/r4y/exchange_timeouts_three_in_flight.txt
