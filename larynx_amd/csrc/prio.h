#pragma once
// Wave priority of the acoustic model's small launches (s_setprio at kernel entry).  Next to the vocoder's MFMA loops a
// young wave otherwise gets the issue slots the older waves leave (priority, then age), and a chain of ~140 dependent
// small launches crawls.  0 = off (A/B builds).
#ifndef MI355TTS_GLOW_PRIO
#define MI355TTS_GLOW_PRIO 3
#endif
#define GLOW_PRIO() do { if (MI355TTS_GLOW_PRIO) __builtin_amdgcn_s_setprio(MI355TTS_GLOW_PRIO); } while (0)
