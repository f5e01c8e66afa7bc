#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "4800 512 512" "19200 512 512"; do
for v in base nt_nomfma nt_nofrag nt_now nt_noa nt_mfmaonly; do
  printf "%-12s " $v; GVL_NT_NW=4 $root/tools/_bin/tgemm_$v nt $shape
done; done
