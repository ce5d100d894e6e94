# Secondary measurements quoted in DESIGN.md (one JSON line each)
python bench.py --height 331 --width 331 --batch 16 --steps 20 2>/dev/null | tail -1 | cut -c1-260
python bench.py --height 331 --width 331 --batch 32 --steps 20 2>/dev/null | tail -1 | cut -c1-260
python bench.py --mode predict --height 331 --width 331 --batch 32 --steps 20 2>/dev/null | tail -1 | cut -c1-260
python bench.py --mode predict --batch 128 --steps 10 2>/dev/null | tail -1 | cut -c1-260
