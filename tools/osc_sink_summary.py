"""profiles/rNN_osc_sink.txt from a bench line:  python tools/osc_sink_summary.py profiles/r06_bench.json > profiles/r06_osc_sink.txt"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
o = d["osc_sink"]
out = ["The OSC sink at scale (bench.py extra `osc_sink`, %s; ref OSCFeatureAnalysisOutput.h:84-113,133, AnalyserTrackController.h:22-23)" % sys.argv[1], "",
       o["loopback"], "bar: " + o["bar"], "",
       "datagrams on the host, ms per tick's worth:   formed on the device (fx_get_osc_datagrams)   |   vectors to the host + fx_osc_encode_batch   (byte-identical)"]
for C in ("1024", "8192", "65536"):
    v = o[C]["datagrams_on_host_ms"]
    out.append("  %6s channels   %.3f ms   |   %.3f ms   (%d B)" % (C, v["device_formed"], v["vectors_then_host_encoder"], v["bytes"]))
out += ["", "sender -> local receiver (as many sockets, SO_REUSEPORT).  back to back = the most a tick loop hands to the kernel; paced = 2 s of the 60 Hz timer with a",
        "fresh publication from the analysis side every tick.  `sustained` = no tick late, nothing dropped, >= 99.9 % received.", "",
        "channels  variant                                      threads  back to back /s   received   paced: ticks  late  datagrams/s   longest tick   sustained 60 Hz"]
names = {"segmented_sends_gro_receiver": "segmented sends, receiver takes them whole", "segmented_sends": "segmented sends, receiver per datagram", "sendmmsg": "sendmmsg, one datagram per message"}
for C in ("1024", "8192", "65536"):
    for n in ("segmented_sends_gro_receiver", "segmented_sends", "sendmmsg"):
        v = o[C][n]; b = v["back_to_back"]; p = v["paced_60hz"]
        out.append("%8s  %-44s %5d  %14.3g   %8.3f   %12d  %4d  %11.4g   %9.2f ms   %s" % (C, names[n], v["threads"], b["handed_to_kernel_per_s"], b["received_share"], p["ticks"], p["late_ticks"],
                                                                                           p["datagrams_per_s"], p["max_tick_ms"], p["sustained"]))
print("\n".join(out))
