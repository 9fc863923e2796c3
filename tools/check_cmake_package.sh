#!/bin/bash
# Configures, builds and installs the CMake package into a scratch prefix, then builds the out-of-tree consumer of
# examples/find_package against it with find_package(yak) (reference yak/CMakeLists.txt:74-153).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=${1:-/tmp/yak_cmake}
rm -rf "$B" && mkdir -p "$B/build" "$B/consumer"
cmake -S "$ROOT" -B "$B/build" -DCMAKE_BUILD_TYPE=Release -DCMAKE_INSTALL_PREFIX="$B/prefix" > "$B/configure.log" 2>&1 || { tail -30 "$B/configure.log"; exit 1; }
cmake --build "$B/build" -j8 > "$B/build.log" 2>&1 || { tail -30 "$B/build.log"; exit 1; }
# the build itself ran the store-data hazard gate on its own assembly of integrate.hip
grep -q "buffer_store_dwordx4 checked, 0 problem" "$B/build.log" && test -f "$B/build/integrate.hazard-checked" \
  || { echo "the CMake build did not run the store-data hazard check"; exit 1; }
echo "store-data hazard gate ran: $(grep 'buffer_store_dwordx4 checked' "$B/build.log")"
cmake --install "$B/build" > "$B/install.log" 2>&1
cmake -S "$ROOT/examples/find_package" -B "$B/consumer" -DCMAKE_PREFIX_PATH="$B/prefix" > "$B/consumer.log" 2>&1 || { tail -30 "$B/consumer.log"; exit 1; }
cmake --build "$B/consumer" >> "$B/consumer.log" 2>&1 || { tail -30 "$B/consumer.log"; exit 1; }
"$B/consumer/consumer"
echo "find_package(yak) consumer built and ran"
