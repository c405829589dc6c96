"""More shape probes (shares check() with check_shapes.py): 96-query tiles at dim 512, 32 query tiles,
fewer rows than queries, dim 1024 with an odd row count."""
import os, sys
sys.path.insert(0, os.getcwd())
sys.argv = ["x"]
exec(open("scripts/check_shapes.py").read().split("check(100_000, 768, 300, 256)")[0])
check(300_000, 512, 1536, 100)      # 96-query tiles at dim 512
check(120_000, 768, 3072, 10)       # 32 tiles of 96
check(64, 768, 1536, 10)            # fewer rows than queries
check(100_003, 1024, 1536, 100)     # dim 1024: 64-query tiles, odd row count
print("adhoc2 ok")
