# same-box A/B of library variants: SETS="setting ..." LIBS="libname ..." (product library = "hip")
set -o pipefail
cd $GRAFT_REPO_ROOT
rm -f gpurun_out/token_time.txt
for rep in 1 2; do
for L in $LIBS; do
  echo "== $L"
  ITTS_HIP_LIB=index-tts-lora_amd/indextts/_lib/libindextts_$L.so timeout -k 10 300 python3 tools/decode_token_time.py $SETS 2>&1 | grep "us/token" | sed "s/^/$L /"
done
done
echo ALLDONE
