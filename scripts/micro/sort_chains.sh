bash scripts/micro/sort_whatif.sh 0 2>&1 | grep -v "rocprofv3\]"
