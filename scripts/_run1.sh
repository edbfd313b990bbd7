python -m pytest tests -m gpu -q -x 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
