#!/bin/bash
# Builds tests/native/sedmi_host_asan: the library's sources + tests/native/host_asan.cpp in one executable with
# AddressSanitizer on the HOST code only (device code is compiled as usual).  Run it on a GPU box:
#   ASAN_OPTIONS=detect_leaks=0 tests/native/sedmi_host_asan
set -e
cd "$(dirname "$0")/.."
SRC=sound_event_detection_amd/csrc
FLAGS="-O1 -g -std=c++17 --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize -Xarch_host -fsanitize=address -Xarch_host -fno-omit-frame-pointer"
mkdir -p /tmp/sedmi_asan
for f in common decode crnn shake; do hipcc $FLAGS -c $SRC/$f.hip -o /tmp/sedmi_asan/$f.o & done
hipcc $FLAGS -fno-signed-zeros -c $SRC/mel.hip -o /tmp/sedmi_asan/mel.o &
hipcc $FLAGS -x hip -c tests/native/host_asan.cpp -o /tmp/sedmi_asan/main.o &
wait
hipcc --offload-arch=gfx950 -Xarch_host -fsanitize=address /tmp/sedmi_asan/*.o -o tests/native/sedmi_host_asan
ls -la tests/native/sedmi_host_asan
