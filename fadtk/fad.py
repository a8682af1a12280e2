from fadtk_amd.fad import *          # noqa: F401,F403
from fadtk_amd.fad import FrechetAudioDistance, calc_embd_statistics, calc_frechet_distance, FADInfResults, log   # noqa: F401
