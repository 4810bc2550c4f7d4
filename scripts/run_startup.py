"""start-up only (302 force evaluations through k_accel*): used under rocprofv3 --stats to time the force kernel"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import ephemeris_explorer_amd as ea
from ephemeris_explorer_amd.workloads import plummer
n = int(sys.argv[1])
pos, vel, mu = plummer(n)
g = ea.NBodyIntegration(pos, vel, mu, 0.0, 1.0 / 1024.0)
g.advance(12)
g.sync()
