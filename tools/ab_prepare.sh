#!/bin/bash
# build container: put the HEAD version of card.io-dmz_amd/csrc/<file> next to the working copy for tools/ab.sh
cd "$(dirname "$0")/.."
git show HEAD:card.io-dmz_amd/csrc/$1 > card.io-dmz_amd/csrc/$1.head
