for f in 0 2 12 6; do echo "FORCE=$f"; SBX_PERMUTE_FORCE_RADIX=$f python tools/permute_probe2.py 2>&1 | grep -v amdgpu.ids | sed 's/digest [0-9a-f]*//'; done
