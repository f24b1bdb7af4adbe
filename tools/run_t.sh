bash tools/ab_so.sh "python tools/attn_one.py 4 20 1024 1024" tools/ab/cur.so tools/ab/attn_w3.so tools/ab/cur.so tools/ab/attn_w3.so
bash tools/ab_so.sh "python tools/attn_one.py 4 10 4096 4096" tools/ab/cur.so tools/ab/attn_w3.so
