#!/bin/bash
# builds the -DKN_KNN_PROF library next to the product one (HERE, before gpurun) and prints the phase split of the screen kernel
set -e
make -C knn_svc_amd/csrc BUILD=build_prof OUT=../libknnsvc_prof.so PROBE=../libknnsvc_prof_probe.so EXTRA="-DKN_QUAD_PROF -DKN_KNN_PROF" -j8 > /dev/null
rm -f knn_svc_amd/libknnsvc_prof_probe.so
