#!/bin/bash
# usage: resusage.sh <build.log> <name-regex>   -- VGPR/scratch/LDS per kernel
awk -v pat="$2" '/Function Name/ {name=$5; show = (name ~ pat)} 
show && /VGPRs:|ScratchSize|LDS Size|VGPRs Spill/ {sub(/.*remark: +/,""); sub(/ \[-Rpass.*/,""); line[name]=line[name]" | "$0}
END {for (n in line) print n, line[n]}' "$1" | sort -u
