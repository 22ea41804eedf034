#!/bin/bash
# build container: put the HEAD version of card.io-dmz_amd/csrc/<file> next to the working copy for tools/ab.sh / tools/ab_bench.sh (second argument: the revision, default HEAD)
cd "$(dirname "$0")/.."
git show ${2:-HEAD}:card.io-dmz_amd/csrc/$1 > card.io-dmz_amd/csrc/$1.head
