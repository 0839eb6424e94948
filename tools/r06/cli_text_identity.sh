#!/bin/bash
# text .sift files of the in-tree libsiftgpu.so against the library of the commit before the round's host-side changes
# (tools/_variants/base_api): same bytes?  (run after tools/r06/cli_list.sh has laid out /tmp/hess_cli_list)
R=${GRAFT_REPO_ROOT:-$PWD}; D=/tmp/hess_cli_list
[ -d $D/a ] || { echo "run tools/r06/cli_list.sh first"; exit 1; }
rm -f $D/a/*.sift $D/b/*.sift
unset LD_LIBRARY_PATH; $R/hessgpu_amd/bin/hess -il $D/a/list.txt -topk 4096 -v 0 > /dev/null 2>&1
LD_LIBRARY_PATH=$R/tools/_variants/base_api $R/hessgpu_amd/bin/hess -il $D/b/list.txt -topk 4096 -v 0 > /dev/null 2>&1
(cd $D/a && md5sum *.sift | awk '{print $1}' | md5sum); (cd $D/b && md5sum *.sift | awk '{print $1}' | md5sum); ls $D/a/*.sift | wc -l; head -c 300 $D/a/img_02_800-1.jpg.sift
