#!/bin/bash
# SURVEY 8(d)'s RMSE protocol (equal seed + converged leg + worst pixels) for BASELINE configs 2 and 3 as well: Cornell all-Diffuse and the material scene, 480 x 270.
set -u
out=gpurun_out/r3rmse; mkdir -p $out
timeout 1500 python tools/rmse_protocol.py --scene cornell_diffuse --size 480x270 --bounces 4 --out $out/rmse_protocol_cornell_diffuse_480x270.json > $out/cornell.log 2>&1; tail -3 $out/cornell.log | cut -c1-200
timeout 2400 python tools/rmse_protocol.py --scene material --size 480x270 --bounces 32 --out $out/rmse_protocol_material_480x270.json > $out/material.log 2>&1; tail -3 $out/material.log | cut -c1-200
