cd /root/repo; O=gpurun_out/r5r; mkdir -p $O
{ for g in 131 262 64 256; do tools/stream_lab $g 1150 3; echo; done; } > $O/stream_lab.txt 2>&1; cat $O/stream_lab.txt
