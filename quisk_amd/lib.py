"""ctypes loader for libquiskhip.so.  Fails loudly when the library is missing.

A process that also uses PyTorch must `import torch` BEFORE the first quisk_amd call: PyTorch ships its own HIP runtime
under the same soname, and whichever copy is loaded first serves both (torch finds no GPU behind /opt/rocm's copy)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QUISKHIP_LIB selects another build of the same library (kernel A/B experiments, tools/ab_bench.py)
LIB_PATH = os.environ.get("QUISKHIP_LIB") or os.path.join(_HERE, "lib", "libquiskhip.so")
_lib = None


class QuiskHipError(RuntimeError):
    pass


def load():
    """Returns the loaded library (ctypes.CDLL) with argument types declared."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QuiskHipError(
            "%s is missing: build it with `python -m quisk_amd.build` (needs hipcc). "
            "quisk_amd has no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i, d, ll = C.c_void_p, C.c_int, C.c_double, C.c_longlong
    L.qh_version.restype = i
    L.qh_last_error.restype = C.c_char_p
    L.qh_device_count.restype = i
    L.qh_rxa_create.restype = vp
    L.qh_rxa_create.argtypes = [i, i, i, i, i, i, vp]
    L.qh_rxa_destroy.argtypes = [vp]
    L.qh_rxa_destroy.restype = None
    for n in ("qh_rxa_nch", "qh_rxa_dsp_insize", "qh_rxa_dsp_outsize"):
        getattr(L, n).argtypes = [vp]
        getattr(L, n).restype = i
    L.qh_rxa_device_bytes.argtypes = [vp]
    L.qh_rxa_device_bytes.restype = ll
    for n in ("SetRXAMode", "RXASetNC", "SetRXAShiftRun", "RXANBPSetRun", "SetRXABandpassRun", "SetRXAAGCMode",
              "SetRXAPanelSelect", "SetRXAPanelCopy", "SetRXAAMDSBMode", "SetRXAAMDFadeLevel", "SetRXACTCSSRun",
              "SetRXAAGCAttack", "SetRXAAGCDecay", "SetRXAAGCHang", "SetRXAAGCSlope", "SetRXAAGCHangThreshold",
              "RXASetMP", "SetRXAAMDRun", "SetRXAFMLimRun", "RXANBPSetNotchesRun", "RXANBPSetWindow", "RXANBPSetAutoIncrease",
              "SetRXAEMNRRun", "SetRXAEMNRgainMethod", "SetRXAEMNRnpeMethod", "SetRXAEMNRaeRun", "SetRXAEMNRPosition", "SetRXASNBARun", "SetRXASNBAasize", "SetRXASNBAnpasses", "SetRXASNBAbridge", "SetRXASNBApresamps", "SetRXASNBApostsamps", "SetRXASNBAovrlp",
              "SetRXAAMSQRun", "SetRXAANFRun", "SetRXAANFTaps", "SetRXAANFDelay", "SetRXAANFPosition", "SetRXAANRRun", "SetRXAANRTaps", "SetRXAANRDelay",
              "SetRXAANRPosition"):
        f = getattr(L, "qh_rxa_" + n)
        f.argtypes = [vp, i, i]
        f.restype = i
    L.qh_rxa_RXANBPAddNotch.argtypes = [vp, i, i, d, d, i, C.POINTER(i)]
    L.qh_rxa_RXANBPEditNotch.argtypes = [vp, i, i, d, d, i, C.POINTER(i)]
    L.qh_rxa_RXANBPDeleteNotch.argtypes = [vp, i, i, C.POINTER(i)]
    L.qh_rxa_RXANBPGetNotch.argtypes = [vp, i, i, C.POINTER(d), C.POINTER(d), C.POINTER(i), C.POINTER(i)]
    L.qh_rxa_RXANBPGetNumNotches.argtypes = [vp, i, C.POINTER(i)]
    L.qh_rxa_RXANBPGetMinNotchWidth.argtypes = [vp, i, C.POINTER(d)]
    for n in ("SetRXAShiftFreq", "SetRXAAGCFixed", "SetRXAPanelGain1", "SetRXAFMDeviation", "SetRXACTCSSFreq", "SetRXAAGCTop",
              "RXANBPSetTuneFrequency", "RXANBPSetShiftFrequency", "SetRXAFMLimGain", "SetRXAAMSQThreshold", "SetRXAAMSQMaxTail", "SetRXAEMNRaeZetaThresh", "SetRXAEMNRaePsi", "SetRXAEMNRtrainZetaThresh", "SetRXASNBAk1", "SetRXASNBAk2", "SetRXASNBApmultmin",
              "SetRXAEMNRtrainT2", "SetRXAANFGain", "SetRXAANFLeakage",
              "SetRXAANRGain", "SetRXAANRLeakage"):
        f = getattr(L, "qh_rxa_" + n)
        f.argtypes = [vp, i, d]
        f.restype = i
    for n in ("RXASetPassband", "RXANBPSetFreqs", "SetRXABandpassFreqs", "SetRXAPanelGain2", "SetRXASNBAOutputBandwidth"):
        f = getattr(L, "qh_rxa_" + n)
        f.argtypes = [vp, i, d, d]
        f.restype = i
    for n in ("SetRXAANFVals", "SetRXAANRVals"):
        getattr(L, "qh_rxa_" + n).argtypes = [vp, i, i, i, d, d]
    L.qh_rxa_SetEMNRTables.argtypes = [vp, vp, vp, vp, vp, d, d, d, d]
    L.qh_rxa_process.argtypes = [vp, vp, ll, vp, ll, i]
    L.qh_rxa_process.restype = i
    L.qh_rxa_process_host.argtypes = [vp, vp, ll, vp, ll, i]
    L.qh_rxa_process_host.restype = i
    L.qh_rxa_synchronize.argtypes = [vp]
    L.qh_rxa_synchronize.restype = i
    L.qh_rxa_enable_meters.argtypes = [vp, i]
    L.qh_rxa_set_graph_replay.argtypes = [vp, i]
    L.qh_rxa_set_band_tile.argtypes = [vp, i]
    L.qh_rxa_band_tile.argtypes = [vp]
    L.qh_rxa_graph_launches.argtypes = [vp]
    L.qh_rxa_graph_launches.restype = ll
    L.qh_rxa_pll_repairs.argtypes = [vp]
    L.qh_rxa_pll_repairs.restype = ll
    L.qh_rxa_debug_agc.argtypes = [vp, i]
    L.qh_rxa_debug_agc.restype = i
    L.qh_rxa_agc_repairs.argtypes = [vp]
    L.qh_rxa_agc_repairs.restype = ll
    L.qh_rxa_agc_segments_rerun.argtypes = [vp]
    L.qh_rxa_agc_segments_rerun.restype = ll
    L.qh_rxa_agc_tiled_channels.argtypes = [vp]
    L.qh_rxa_agc_tiled_channels.restype = i
    L.qh_rxa_process_audio.argtypes = [vp, vp, ll, vp, ll, i, vp]
    L.qh_rxa_process_audio.restype = i
    L.qh_audio_pack.argtypes = [i, vp, vp, ll, i, i, vp, vp, ll]
    L.qh_audio_pack.restype = i
    L.qh_wdsp_graph_launches.restype = ll
    L.qh_wdsp_fexchange0_device.argtypes = [i, vp, i, vp]
    L.qh_wdsp_fexchange0_device.restype = i
    L.qh_rxa_GetRXAMeter.argtypes = [vp, i, i, C.POINTER(d)]
    L.qh_rxa_flush.argtypes = [vp]
    L.qh_rxa_flush.restype = i
    L.qh_rxa_enable_timing.argtypes = [vp, i]
    L.qh_rxa_enable_timing.restype = i
    L.qh_rxa_timing.argtypes = [vp, C.POINTER(d), i]
    L.qh_rxa_timing.restype = i
    L.qh_fir_create.restype = vp
    L.qh_fir_create.argtypes = [i, i, vp, vp, i, i, i, vp]
    L.qh_fir_destroy.argtypes = [vp]
    L.qh_fir_destroy.restype = None
    L.qh_fir_reset.argtypes = [vp]
    L.qh_fir_out_count.argtypes = [vp, i]
    L.qh_fir_process.argtypes = [vp, vp, ll, i, vp, ll, C.POINTER(i)]
    L.qh_fir_process_host.argtypes = [vp, vp, ll, i, vp, ll, C.POINTER(i)]
    L.qh_fir_synchronize.argtypes = [vp]
    L.qh_rat_create.restype = vp
    L.qh_rat_create.argtypes = [i, i, vp, i, i, i, i, vp]
    L.qh_rat_destroy.argtypes = [vp]
    L.qh_rat_destroy.restype = None
    L.qh_rat_reset.argtypes = [vp]
    L.qh_rat_set_state.argtypes = [vp, vp, i]
    L.qh_rat_phase.argtypes = [vp]
    L.qh_rat_out_count.argtypes = [vp, i]
    L.qh_rat_process.argtypes = [vp, vp, ll, i, vp, ll, C.POINTER(i)]
    L.qh_rat_process_host.argtypes = [vp, vp, ll, i, vp, ll, C.POINTER(i)]
    L.qh_rat_synchronize.argtypes = [vp]
    L.qh_hbc_create.restype = vp
    L.qh_hbc_create.argtypes = [i, i, i, i, vp]
    L.qh_hbc_destroy.argtypes = [vp]
    L.qh_hbc_destroy.restype = None
    L.qh_hbc_reset.argtypes = [vp]
    L.qh_hbc_process.argtypes = [vp, vp, ll, i, vp, ll]
    L.qh_hbc_process_host.argtypes = [vp, vp, ll, i, vp, ll]
    L.qh_hbc_synchronize.argtypes = [vp]
    L.qh_hb45_taps.argtypes = [vp]
    L.qh_hb45_taps.restype = None
    L.qh_pan_create.restype = vp
    L.qh_pan_create.argtypes = [i, i, i, i, d, vp]
    L.qh_pan_destroy.argtypes = [vp]
    L.qh_pan_destroy.restype = None
    L.qh_pan_set_smeter_band.argtypes = [vp, i, d, d]
    L.qh_pan_feed.argtypes = [vp, vp, ll, i]
    L.qh_pan_feed_host.argtypes = [vp, vp, ll, i]
    L.qh_pan_count.argtypes = [vp]
    L.qh_pan_graph.argtypes = [vp, d, d, vp, vp, C.POINTER(i)]
    L.qh_pan_waterfall.argtypes = [vp, d, d, vp, vp, vp, i, i, d, i, vp, vp, C.POINTER(i)]
    L.qh_watfall_rows_host.argtypes = [i, vp, i, i, vp, vp, vp, i, i, d, i, vp]
    L.qh_bscope_create.restype = vp
    L.qh_bscope_create.argtypes = [i, i, i, i, vp]
    L.qh_bscope_destroy.argtypes = [vp]
    L.qh_bscope_destroy.restype = None
    L.qh_bscope_feed.argtypes = [vp, vp, ll, i]
    L.qh_bscope_feed_host.argtypes = [vp, vp, ll, i]
    L.qh_bscope_count.argtypes = [vp]
    L.qh_bscope_graph.argtypes = [vp, i, d, d, vp, vp, C.POINTER(i)]
    L.qh_qrx_create.restype = vp
    L.qh_qrx_create.argtypes = [i, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp]
    L.qh_iq_format_le24.argtypes = [vp, C.c_double]
    L.qh_iq_format_le24.restype = None
    L.qh_iq_format_hermes.argtypes = [vp, i, C.c_double]
    L.qh_iq_format_hermes.restype = None
    L.qh_unpack_udp17.argtypes = [i, vp, vp, i, i, d, i, d, d, vp, vp, vp, vp, vp]
    L.qh_unpack_udp17_host.argtypes = [i, vp, i, i, d, i, d, d, vp, vp, vp, vp, vp]
    L.qh_unpack_iq.argtypes = [i, vp, vp, ll, vp, i, ll, i, vp, ll, i]
    L.qh_unpack_iq_host.argtypes = [i, vp, ll, vp, i, ll, i, vp, ll, i]
    L.qh_rxa_process_packed.argtypes = [vp, vp, ll, vp, ll, vp, ll, i]
    L.qh_rxa_process_packed_host.argtypes = [vp, vp, ll, vp, ll, vp, ll, i]
    L.qh_qagc_create.restype = vp
    L.qh_qagc_create.argtypes = [i, i, i, C.c_double, C.c_double, i, vp]
    L.qh_qagc_destroy.argtypes = [vp]
    L.qh_qagc_destroy.restype = None
    L.qh_qagc_set_gain.argtypes = [vp, i, C.c_double]
    L.qh_qagc_reset.argtypes = [vp]
    L.qh_qagc_set_cpx.argtypes = [vp, i]
    L.qh_qagc_process.argtypes = [vp, vp, ll, i]
    L.qh_qagc_process2.argtypes = [vp, vp, ll, vp, ll, i]
    L.qh_qagc_debug_form.argtypes = [vp, i]
    L.qh_qagc_process_host.argtypes = [vp, vp, ll, i]
    L.qh_ana_create.restype = vp
    L.qh_ana_create.argtypes = [i, i, i, i, vp]
    L.qh_ana_destroy.argtypes = [vp]
    L.qh_ana_destroy.restype = None
    L.qh_ana_set_analyzer.argtypes = [vp, i, i, i, C.POINTER(i), i, i, i, d, i, i, d, d, i, i, i, d, d, i]
    L.qh_ana_set_calibration.argtypes = [vp, i, i, vp]
    for n in ("detector_mode", "average_mode", "num_average", "norm_onehz"):
        getattr(L, "qh_ana_set_" + n).argtypes = [vp, i, i]
    L.qh_ana_set_av_backmult.argtypes = [vp, i, d]
    L.qh_ana_set_sample_rate.argtypes = [vp, i]
    L.qh_ana_get_enb.argtypes = [vp]
    L.qh_ana_get_enb.restype = d
    L.qh_ana_reset_pixel_buffers.argtypes = [vp]
    L.qh_ana_feed.argtypes = [vp, i, vp, ll, i, C.POINTER(i)]
    L.qh_ana_feed_host.argtypes = [vp, i, vp, ll, i, i, C.POINTER(i)]
    L.qh_ana_get_pixels.argtypes = [vp, i, i, vp, C.POINTER(i)]
    L.qh_ana_rows.argtypes = [vp, i, C.POINTER(vp), C.POINTER(i), C.POINTER(i)]
    L.qh_ana_rows_host.argtypes = [vp, i, vp, i, C.POINTER(i)]
    L.qh_ana_stream.argtypes = [vp]
    L.qh_ana_stream.restype = vp
    L.qh_ana_frames.argtypes = [vp]
    L.qh_ana_frames.restype = ll
    L.XCreateAnalyzer.argtypes = [i, C.POINTER(i), i, i, i, C.c_char_p]
    L.XCreateAnalyzer.restype = None
    L.DestroyAnalyzer.argtypes = [i]
    L.DestroyAnalyzer.restype = None
    L.SetAnalyzer.argtypes = [i, i, i, i, C.POINTER(i), i, i, i, d, i, i, d, d, i, i, i, d, d, i]
    L.SetAnalyzer.restype = None
    L.Spectrum0.argtypes = [i, i, i, i, vp]
    L.Spectrum0.restype = None
    L.Spectrum2.argtypes = [i, i, i, i, vp]
    L.Spectrum2.restype = None
    L.Spectrum.argtypes = [i, i, i, vp, vp]
    L.Spectrum.restype = None
    L.OpenBuffer.argtypes = [i, i, i, C.POINTER(vp), C.POINTER(vp)]
    L.OpenBuffer.restype = None
    L.CloseBuffer.argtypes = [i, i, i]
    L.CloseBuffer.restype = None
    L.GetPixels.argtypes = [i, i, vp, C.POINTER(i)]
    L.GetPixels.restype = None
    L.SetCalibration.argtypes = [i, i, i, vp]
    L.SetCalibration.restype = None
    L.ResetPixelBuffers.argtypes = [i]
    L.ResetPixelBuffers.restype = None
    for n in ("SetDisplayDetectorMode", "SetDisplayAverageMode", "SetDisplayNumAverage", "SetDisplayNormOneHz"):
        getattr(L, n).argtypes = [i, i, i]
        getattr(L, n).restype = None
    L.SetDisplayAvBackmult.argtypes = [i, i, d]
    L.SetDisplayAvBackmult.restype = None
    L.SetDisplaySampleRate.argtypes = [i, i]
    L.SetDisplaySampleRate.restype = None
    L.GetDisplayENB.argtypes = [i]
    L.GetDisplayENB.restype = d
    L.qh_nb_create.restype = vp
    L.qh_nb_create.argtypes = [i, i, i, vp]
    L.qh_nb_destroy.argtypes = [vp]
    L.qh_nb_destroy.restype = None
    L.qh_nb_delay.argtypes = [vp]
    L.qh_nb_set_level.argtypes = [vp, i]
    L.qh_nb_reset.argtypes = [vp]
    L.qh_nb_process.argtypes = [vp, vp, ll, vp, ll, i]
    L.qh_nb_process_host.argtypes = [vp, vp, ll, vp, ll, i]
    L.qh_nb_synchronize.argtypes = [vp]
    L.qh_qrx_set_noise_blanker.argtypes = [vp, i]
    L.qh_qrx_set_auto_notch.argtypes = [vp, i, i]
    L.qh_quisk_set_auto_notch.argtypes = [i, i]
    L.qh_quisk_set_auto_notch.restype = None
    L.qh_quisk_set_noise_blanker.argtypes = [i]
    L.qh_quisk_set_noise_blanker.restype = None
    L.qh_qrx_set_agc.argtypes = [vp, i, C.c_double]
    L.qh_qrx_set_squelch.argtypes = [vp, i, C.c_double]
    L.qh_qrx_set_ssb_squelch.argtypes = [vp, i, i]
    L.qh_quisk_open.argtypes = [i, i, vp, i, i]
    L.qh_quisk_close.restype = None
    L.qh_quisk_set_tune.argtypes = [i]
    L.qh_quisk_set_tune.restype = None
    L.qh_quisk_set_rx_mode.argtypes = [i]
    L.qh_quisk_set_rx_mode.restype = None
    L.qh_quisk_set_filters.argtypes = [vp, vp, i, i]
    L.qh_quisk_set_agc.argtypes = [C.c_double]
    L.qh_quisk_set_agc.restype = None
    L.qh_quisk_process_samples.argtypes = [vp, i]
    for n, at in (("set_tx_tune", [i]), ("set_split_rxtx", [i]), ("set_multirx_play_channel", [i]), ("set_multirx_play_method", [i]),
                  ("set_multirx_freq", [i, i]), ("set_multirx_mode", [i, i]), ("set_key_state", [i, i, i, i]),
                  ("set_sidetone", [C.c_double, i, i, i]), ("set_kill_audio", [i]), ("invert_spectrum", [i]),
                  ("set_squelch", [C.c_double]), ("set_ssb_squelch", [i, i]), ("add_tone", [i]), ("set_multirx_count", [i]),
                  ("set_sub_rx1_output", [i])):
        f = getattr(L, "qh_quisk_" + n)
        f.argtypes = at
        f.restype = None
    L.qh_quisk_multirx_samples.argtypes = [i, vp, i]
    L.qh_quisk_set_filters2.argtypes = [vp, vp, i, i]
    L.qh_quisk_set_filters_n.argtypes = [vp, vp, i, i, i]
    L.qh_pan_attach_fir.argtypes = [vp, vp, i, i]
    L.qh_pan_feed_decimate.argtypes = [vp, vp, ll, i, vp, ll, vp]
    L.qh_quisk_measure_frequency.argtypes = [i]
    L.qh_quisk_measure_frequency.restype = C.c_double
    L.qh_quisk_sub_rx1_audio.argtypes = [vp, i]
    L.qh_quisk_squelch_flags.argtypes = []
    L.qh_quisk_get_graph.argtypes = [C.c_double, C.c_double, vp, vp]
    L.qh_ana_snap_arm.argtypes = [vp, i, i]
    L.qh_qps_create.restype = vp
    L.qh_qps_create.argtypes = [i, i, i, i, i, i, vp, i, i, vp]
    L.qh_qps_destroy.argtypes = [vp]
    L.qh_qps_destroy.restype = None
    L.qh_qps_set_tune.argtypes = [vp, i, i]
    L.qh_qps_set_tune_all.argtypes = [vp, vp]
    L.qh_qps_set_filters.argtypes = [vp, i, vp, vp, i]
    L.qh_qps_set_agc.argtypes = [vp, d]
    for n in ("qh_qps_set_noise_blanker", "qh_qps_invert_spectrum", "qh_qps_set_kill_audio", "qh_qps_add_tone", "qh_qps_set_pieces", "qh_qps_set_pipelined"):
        getattr(L, n).argtypes = [vp, i]
    L.qh_qps_set_auto_notch.argtypes = [vp, i, i]
    L.qh_qps_set_squelch.argtypes = [vp, i, d]
    L.qh_qps_set_ssb_squelch.argtypes = [vp, i, i]
    for n in ("qh_qps_filter_rate", "qh_qps_decim_rate", "qh_qps_synchronize"):
        getattr(L, n).argtypes = [vp]
    L.qh_qps_out_capacity.argtypes = [vp, i]
    L.qh_qps_process.argtypes = [vp, vp, ll, i, vp, ll, C.POINTER(i)]
    L.qh_qps_process_host.argtypes = [vp, vp, ll, i, vp, ll, C.POINTER(i)]
    L.qh_qps_squelch_flags.argtypes = [vp, vp]
    L.qh_qps_get_graph.argtypes = [vp, d, d, vp, vp, C.POINTER(i)]
    L.qh_quisk_error_count.restype = ll
    L.qh_qrx_create_ex.restype = vp
    L.qh_qrx_create_ex.argtypes = [i, i, i, i, i, vp, vp]
    L.qh_qrx_decim_rate.argtypes = [vp]
    L.qh_qrx_destroy.argtypes = [vp]
    L.qh_qrx_destroy.restype = None
    L.qh_qrx_filter_rate.argtypes = [vp]
    L.qh_qrx_set_tune.argtypes = [vp, i, i]
    L.qh_qrx_set_tune_all.argtypes = [vp, vp]
    L.qh_qrx_set_filters.argtypes = [vp, i, vp, vp, i]
    L.qh_qrx_out_count.argtypes = [vp, i]
    L.qh_qrx_process.argtypes = [vp, vp, ll, i, vp, ll, C.POINTER(i)]
    L.qh_qrx_process_host.argtypes = [vp, vp, ll, i, vp, ll, C.POINTER(i)]
    L.qh_qrx_synchronize.argtypes = [vp]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise QuiskHipError("libquiskhip error %d: %s" % (rc, load().qh_last_error().decode(errors="replace")))
