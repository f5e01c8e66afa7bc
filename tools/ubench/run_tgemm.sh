#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}
for shape in "4800 512 512" "19200 512 512" "2208 8520 512"; do
  for v in base nomfma nofrag nostore noload mfmaonly; do
    printf "%-9s " $v; $root/tools/_bin/tgemm_$v wgrad $shape
  done
done
