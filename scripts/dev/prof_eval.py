import cProfile, pstats, sys, io
sys.path.insert(0, '.')
sys.argv = ['evaluate_real.py', '--synthetic', '--max_sequences', '6']
import runpy
pr = cProfile.Profile()
pr.enable()
try:
    runpy.run_path('scripts/evaluate_real.py', run_name='__main__')
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[-5000:])
