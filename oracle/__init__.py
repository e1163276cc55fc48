"""TEST INFRASTRUCTURE — CPU oracle of the Larynx hot path (phoneme ids ->
GlowTTS -> mel transform -> HiFi-GAN -> waveform).  Never imported by the
product package `larynx_amd`; see oracle/README.md."""
