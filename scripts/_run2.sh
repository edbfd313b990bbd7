python scripts/_dbg_join.py c2_b2 2>&1 | tail -30
echo ---- f32 hidden
DG_HIDDEN=f32 python scripts/_dbg_join.py c2_b2 2>&1 | tail -12
echo ---- forward traversal
DG_TRAVERSAL=forward python scripts/_dbg_join.py c2_b2 2>&1 | tail -12
