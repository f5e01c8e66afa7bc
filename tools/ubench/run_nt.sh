#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "4800 512 512"; do
for v in base nt_nomfma nt_mfmaonly; do
  printf "%-12s " $v; GVL_NT_NW=4 $root/tools/_bin/tgemm_$v nt $shape
done; done
for nw in 4 3 2; do for shape in "4800 512 512" "3008 512 512" "4800 1536 512" "4800 256 512" "3008 1536 512"; do printf "NW=$nw "; GVL_NT_NW=$nw $root/tools/_bin/tgemm_base nt $shape | head -1; done; done
