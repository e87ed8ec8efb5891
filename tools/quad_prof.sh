#!/bin/bash
# builds the -DKN_QUAD_PROF library next to the product one and prints the phase split of the quad kernel on the encoder's shapes
set -e
[ -f knn_svc_amd/libknnsvc_prof.so ] || make -C knn_svc_amd/csrc BUILD=build_prof OUT=../libknnsvc_prof.so PROBE=../libknnsvc_prof_probe.so EXTRA="-DKN_QUAD_PROF -DKN_KNN_PROF" -j8 > /dev/null
export KNNSVC_LIB=$PWD/knn_svc_amd/libknnsvc_prof.so KNNSVC_QUADP=0 KNNSVC_QUAD=2
ACT=gelu OSPLIT=1 python tools/quad_prof.py 31500 4096 1024
python tools/quad_prof.py 31500 3072 1024
RESID=1 python tools/quad_prof.py 31500 1024 1024
RESID=1 python tools/quad_prof.py 31500 1024 4096
