#!/bin/bash
# hess -il over a list of JPEG files: wall time per image with the decode-ahead (in-tree libsiftgpu.so) and without (the
# library of the commit before, tools/_variants/base_api): tools/r06/cli_list.sh
R=${GRAFT_REPO_ROOT:-$PWD}
D=/tmp/hess_cli_list; rm -rf $D; mkdir -p $D/a $D/b
for d in a b; do
  i=0
  for rep in 1 2 3 4; do for f in 1600.jpg 800-1.jpg 800-2.jpg 800-3.jpg 800-4.jpg 640-1.jpg 640-2.jpg 640-3.jpg 640-4.jpg 640-5.jpg; do
    i=$((i+1)); cp $R/tests/golden/data/$f $D/$d/img_$(printf %02d $i)_$f; done; done
  (cd $D/$d && ls *.jpg > list.txt)
done
n=$(wc -l < $D/a/list.txt)
for rnd in 1 2; do
for mode in base new; do
  if [ "$mode" = base ]; then dir=$D/b; export LD_LIBRARY_PATH=$R/tools/_variants/base_api; else dir=$D/a; unset LD_LIBRARY_PATH; fi
  for fmt in "" "-b"; do
    rm -f $dir/*.sift
    t0=$(date +%s.%N)
    $R/hessgpu_amd/bin/hess -il $dir/list.txt -topk 4096 $fmt -v 0 > /dev/null 2> $D/err_$mode.txt || { echo "hess failed ($mode)"; tail -3 $D/err_$mode.txt; exit 9; }
    t1=$(date +%s.%N)
    echo "$mode ${fmt:-text} : $(python3 -c "print(f'{($t1-$t0)*1e3/$n:.2f} ms per image ({$n} images, {($t1-$t0):.2f} s)')")"
  done
done
done
# the .sift files of the two builds (binary format, last pass) are the same bytes
(cd $D/a && md5sum *.sift | awk '{print $1}' | md5sum) ; (cd $D/b && md5sum *.sift | awk '{print $1}' | md5sum)
# marginal cost per image (process start-up and context creation cancel): 120 images against 40, binary output
(cd $D/a && for k in 1 2; do for f in img_*.jpg; do cp $f x${k}_$f; done; done && ls *.jpg > list3.txt)
for mode in new; do
  unset LD_LIBRARY_PATH
  t0=$(date +%s.%N); $R/hessgpu_amd/bin/hess -il $D/a/list.txt -topk 4096 -b -v 0 > /dev/null 2>&1; t1=$(date +%s.%N)
  $R/hessgpu_amd/bin/hess -il $D/a/list3.txt -topk 4096 -b -v 0 > /dev/null 2>&1; t2=$(date +%s.%N)
  python3 -c "print(f'marginal, -b: {(($t2-$t1)-($t1-$t0))*1e3/80:.2f} ms per image  (40 images {($t1-$t0):.2f} s, 120 images {($t2-$t1):.2f} s)')"
  t0=$(date +%s.%N); $R/hessgpu_amd/bin/hess -il $D/a/list.txt -topk 4096 -v 0 > /dev/null 2>&1; t1=$(date +%s.%N)
  $R/hessgpu_amd/bin/hess -il $D/a/list3.txt -topk 4096 -v 0 > /dev/null 2>&1; t2=$(date +%s.%N)
  python3 -c "print(f'marginal, text: {(($t2-$t1)-($t1-$t0))*1e3/80:.2f} ms per image  (40 images {($t1-$t0):.2f} s, 120 images {($t2-$t1):.2f} s)')"
done
export LD_LIBRARY_PATH=$R/tools/_variants/base_api
(cd $D/b && for k in 1 2; do for f in img_*.jpg; do cp $f x${k}_$f; done; done && ls *.jpg > list3.txt)
t0=$(date +%s.%N); $R/hessgpu_amd/bin/hess -il $D/b/list.txt -topk 4096 -b -v 0 > /dev/null 2>&1; t1=$(date +%s.%N)
$R/hessgpu_amd/bin/hess -il $D/b/list3.txt -topk 4096 -b -v 0 > /dev/null 2>&1; t2=$(date +%s.%N)
python3 -c "print(f'marginal, -b, library before: {(($t2-$t1)-($t1-$t0))*1e3/80:.2f} ms per image')"
unset LD_LIBRARY_PATH
(cd $D/a && rm -f x?_*.jpg x?_*.sift list3.txt); (cd $D/b && rm -f x?_*.jpg x?_*.sift list3.txt)
# decode alone, one thread (the debug hook decodes as RunSIFT(path) would)
python3 - <<PY
import ctypes, time, glob, os
L = ctypes.CDLL("$R/hessgpu_amd/libsiftgpu.so")
L.siftgpu_debug_load_image.argtypes = [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
files = sorted(glob.glob("$D/a/*.jpg"))
w, h = ctypes.c_int(), ctypes.c_int()
t0 = time.perf_counter()
for f in files:
    L.siftgpu_debug_load_image(f.encode(), None, 0, ctypes.byref(w), ctypes.byref(h))
print(f"decode alone, one thread: {(time.perf_counter() - t0) * 1e3 / len(files):.2f} ms per image")
PY
