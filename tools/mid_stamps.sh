#!/bin/bash
# k_mid by block kind (round 6): cycle stamps of a -DMID_STAMPS build (tools/bin/libtsba_midstamps.so) put in place of the product library for this run only
set -u
cd "${GRAFT_REPO_ROOT:-.}"
cp textslam_amd/libtsba.so /tmp/libtsba_prod.so; cp tools/bin/libtsba_midstamps.so textslam_amd/libtsba.so
python tools/diag/gpu_mid_stamps.py
cp /tmp/libtsba_prod.so textslam_amd/libtsba.so
