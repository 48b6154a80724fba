#!/usr/bin/env bash
# usage: bash tools/stamp_isa.sh   — the hash tools/isa_stats.sh stamps its record with (tests/test_abi_and_host.py compares it with the tree)
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
echo "# sources sha256: $(cd "$HERE/python-wlsqm_amd" && export LC_ALL=C && sha256sum csrc/*.hip csrc/*.hpp ../include/*.h | sed 's#\.\./include#include#' | sha256sum | cut -d' ' -f1)"
