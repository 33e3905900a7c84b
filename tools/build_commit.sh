#!/bin/bash
# Build the library as of an earlier commit, for an in-box A/B against the working tree:
#   tools/build_commit.sh <tag> <commit>   ->  rtl-wmbus_amd/libwmbus_hip_<tag>.so
# (the GPU box gets a snapshot without .git, so the "before" build has to be made here and travels with it)
set -e
tag=$1; commit=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/wmbus_build_XXXXXX)
git -C "$root" archive "$commit" rtl-wmbus_amd include | tar -x -C "$tmp"
make -s -C "$tmp/rtl-wmbus_amd" "$tmp/rtl-wmbus_amd/libwmbus_hip.so" >/dev/null
cp "$tmp/rtl-wmbus_amd/libwmbus_hip.so" "$root/rtl-wmbus_amd/libwmbus_hip_$tag.so"
rm -rf "$tmp"
ls -la "$root/rtl-wmbus_amd/libwmbus_hip_$tag.so"
