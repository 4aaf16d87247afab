/* gstpeaq_amd.c -- the `peaq` GStreamer element on top of libpeaq_amd.so.
 *
 * Host side of the drop-in boundary: same element name, pads,
 * caps, properties, console text and flush semantics as the element of
 * HSU-ANT/gstpeaq (reference src/gstpeaq.c), but everything below the adapters
 * -- ear models, pattern processing, MOVs, accumulators, neural network -- is
 * one peaq_session of the MI355X engine (include/peaq_amd.h).
 *
 *   element "peaq", klass Sink/Audio, GST_ELEMENT_FLAG_SINK   gstpeaq.c:319-323,355
 *   sink pads "ref" and "test", ALWAYS                         gstpeaq.c:154-165
 *   caps audio/x-raw F32LE interleaved 48000 Hz                gstpeaq.c:146-152
 *   both pads negotiate the same channel count                 gstpeaq.c:216-244,689-708
 *   properties playback_level, advanced, console-output,
 *              di, odg, totalsnr                               gstpeaq.c:273-317
 *   PAUSED->READY flushes with zero padding and evaluates ODG  gstpeaq.c:764-778
 */
#include <gst/gst.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "peaq_amd.h"

#ifndef PACKAGE
#define PACKAGE "gstpeaq-amd"
#endif
#ifndef PACKAGE_VERSION
#define PACKAGE_VERSION "0.2.0"
#endif

GST_DEBUG_CATEGORY_STATIC (peaq_amd_debug);
#define GST_CAT_DEFAULT peaq_amd_debug

#define PEAQ_CAPS "audio/x-raw, format = (string) F32LE, layout = (string) interleaved, rate = (int) 48000"

typedef struct _GstPeaqAmd
{
  GstElement element;
  GstPad *pads[2];              /* 0 = ref, 1 = test */
  gboolean eos[2];
  gboolean advanced;
  gboolean console_output;
  gdouble playback_level;
  gint channels;
  peaq_session *session;        /* NULL until caps are known */
  peaq_broker *broker;          /* set instead of `session` when the process-wide broker serves this element */
  gint broker_sid;
  gboolean failed;              /* a device error was reported already */
  gboolean pushed;              /* the current session / broker slot has received samples */
  gchar *pending_error;         /* set under the object lock, posted after it is released */
} GstPeaqAmd;

typedef struct _GstPeaqAmdClass
{
  GstElementClass parent_class;
} GstPeaqAmdClass;

enum
{ PROP_0, PROP_PLAYBACK_LEVEL, PROP_ADVANCED, PROP_DI, PROP_ODG, PROP_TOTALSNR, PROP_CONSOLE_OUTPUT };

static GstStaticPadTemplate ref_template =
GST_STATIC_PAD_TEMPLATE ("ref", GST_PAD_SINK, GST_PAD_ALWAYS, GST_STATIC_CAPS (PEAQ_CAPS));
static GstStaticPadTemplate test_template =
GST_STATIC_PAD_TEMPLATE ("test", GST_PAD_SINK, GST_PAD_ALWAYS, GST_STATIC_CAPS (PEAQ_CAPS));

/* registered under the reference's type name "GstPeaq" so that unnamed instances
 * are called peaq0, peaq1, ... as before */
typedef GstPeaqAmd GstPeaq;
typedef GstPeaqAmdClass GstPeaqClass;
GType gst_peaq_amd_get_type (void);
G_DEFINE_TYPE_WITH_CODE (GstPeaq, gst_peaq_amd, GST_TYPE_ELEMENT,);
#define GST_PEAQ_AMD(obj) ((GstPeaqAmd *) (obj))

/* one device context per process */
static peaq_ctx *
shared_context (void)
{
  static gsize once = 0;
  static peaq_ctx *ctx = NULL;
  if (g_once_init_enter (&once)) {
    const gchar *dev = g_getenv ("PEAQ_AMD_DEVICE");
    if (peaq_ctx_create (dev ? atoi (dev) : 0, &ctx) != PEAQ_OK) {
      GST_ERROR ("libpeaq_amd: %s", peaq_last_error ());
      ctx = NULL;
    } else {
      /* The advanced version's filter bank runs in the engine's default arithmetic, the reference's own (all
       * FP64), whether the elements keep their own sessions or share a broker; PEAQ_AMD_FIR=f16x3 in the
       * environment selects the faster reduced-precision bank (peaq_ctx_create reads it). */
      /* decided once per process, when the first element is created: say which it was */
      GST_INFO ("libpeaq_amd: device %d, filter-bank arithmetic of the advanced version: %s", peaq_ctx_device (ctx),
                peaq_ctx_get_fir_mode (ctx) == PEAQ_FIR_F64 ? "f64" : peaq_ctx_get_fir_mode (ctx) == PEAQ_FIR_F32 ? "f32" : "f16x3");
    }
    g_once_init_leave (&once, 1);
  }
  return ctx;
}

/* PEAQ_AMD_BROKER=<max sessions>: a process that hosts many `peaq` elements lets ONE broker run
 * the frames of all of them as one batched launch per tick (include/peaq_amd.h, "broker").  One
 * broker per (version, channel count) at the default playback level; an element with another
 * level keeps its own session.  PEAQ_AMD_DEVICES=0,1,...: the broker spans those GPUs
 * (peaq_broker_create_multi: one device broker each, elements dealt out to the least loaded). */
static peaq_broker *
shared_broker (gboolean advanced, gint channels)
{
  static GMutex lock;
  static peaq_broker *brokers[2][3] = { {NULL, NULL, NULL}, {NULL, NULL, NULL} };
  const gchar *max = g_getenv ("PEAQ_AMD_BROKER");
  const gint adv = advanced ? 1 : 0;
  peaq_broker *b = NULL;
  if (!max || atoi (max) <= 0 || channels < 1 || channels > 2)
    return NULL;
  g_mutex_lock (&lock);
  if (!brokers[adv][channels]) {
    const gchar *period = g_getenv ("PEAQ_AMD_BROKER_PERIOD_US");
    const gchar *devs = g_getenv ("PEAQ_AMD_DEVICES");
    gint devices[64], n_devices = 0, rc = PEAQ_OK;
    if (devs) {
      /* "0,1,3": device ordinals, nothing else -- a token that is not a non-negative number is an error, not device 0 */
      gchar **tok = g_strsplit (devs, ",", 64);
      for (gint i = 0; tok[i] && n_devices < 64 && rc == PEAQ_OK; i++) {
        gchar *t = g_strstrip (tok[i]), *end = NULL;
        gint64 v;
        if (!*t)
          continue;
        v = g_ascii_strtoll (t, &end, 10);
        if (end == t || *end || v < 0 || v > 1023) {
          GST_WARNING ("PEAQ_AMD_DEVICES: '%s' is not a device ordinal; no broker, one session per element", t);
          rc = PEAQ_ERR_ARG;
        } else
          devices[n_devices++] = (gint) v;
      }
      g_strfreev (tok);
    }
    if (rc != PEAQ_OK) {
      g_mutex_unlock (&lock);
      return NULL;
    }
    if (n_devices > 0) {
      /* one context per listed device inside the broker; the process-wide context of device 0 is not needed (fir_mode
       * -1: the engine's default and the environment apply to every device's context alike) */
      rc = peaq_broker_create_multi (devices, n_devices, adv, channels, 92., MAX (atoi (max), n_devices), NULL, -1,
                                     &brokers[adv][channels]);
    } else {
      peaq_ctx *ctx = shared_context ();
      rc = ctx ? peaq_broker_create (ctx, adv, channels, 92., atoi (max), &brokers[adv][channels]) : PEAQ_ERR_DEVICE;
    }
    if (rc == PEAQ_OK) {
      if (peaq_broker_start (brokers[adv][channels], period ? (unsigned) atoi (period) : 0) != PEAQ_OK)
        GST_WARNING ("libpeaq_amd: %s", peaq_last_error ());
    } else {
      GST_WARNING ("libpeaq_amd: no broker (%s), using one session per element", peaq_last_error ());
      brokers[adv][channels] = NULL;
    }
  }
  b = brokers[adv][channels];
  g_mutex_unlock (&lock);
  return b;
}

static void
drop_session (GstPeaqAmd * self)
{
  if (self->session)
    peaq_session_destroy (self->session);
  if (self->broker)
    peaq_broker_close (self->broker, self->broker_sid);
  self->session = NULL;
  self->broker = NULL;
}

/* (re)create the engine session: the reference re-allocates all per-channel state
 * whenever caps or the `advanced` property change (gstpeaq.c:519,559,575,586).
 * Called with the object lock held: a failure is only RECORDED here
 * (pending_error); GST_ELEMENT_ERROR takes the same non-recursive lock, so the
 * callers post it through post_pending_error() after unlocking. */
static gboolean
renew_session (GstPeaqAmd * self)
{
  peaq_ctx *ctx;
  drop_session (self);
  self->pushed = FALSE;
  if (self->channels <= 0)
    return TRUE;
  if (self->playback_level == 92.) {
    peaq_broker *b = shared_broker (self->advanced, self->channels);
    if (b && peaq_broker_open (b, &self->broker_sid) == PEAQ_OK) {
      self->broker = b;
      return TRUE;
    }
  }
  ctx = shared_context ();
  if (!ctx || peaq_session_create (ctx, self->advanced, self->channels, self->playback_level,
          &self->session) != PEAQ_OK) {
    g_free (self->pending_error);
    self->pending_error = g_strdup_printf ("libpeaq_amd: %s", peaq_last_error ());
    self->session = NULL;
    return FALSE;
  }
  return TRUE;
}

/* call WITHOUT the object lock */
static void
post_pending_error (GstPeaqAmd * self)
{
  gchar *msg;
  GST_OBJECT_LOCK (self);
  msg = self->pending_error;
  self->pending_error = NULL;
  GST_OBJECT_UNLOCK (self);
  if (msg) {
    GST_ELEMENT_ERROR (self, LIBRARY, INIT, ("%s", msg), (NULL));
    g_free (msg);
  }
}

static gboolean
read_results (GstPeaqAmd * self, peaq_result * r)
{
  if (self->broker && peaq_broker_results (self->broker, self->broker_sid, r) == PEAQ_OK)
    return TRUE;
  if (!self->session || peaq_session_results (self->session, r) != PEAQ_OK) {
    gint i;
    for (i = 0; i < PEAQ_MOVS_BASIC; i++)
      r->movs[i] = NAN;
    r->di = r->odg = r->totalsnr = NAN;      /* no data yet: the reference's empty accumulators give NaN too */
    return FALSE;
  }
  return TRUE;
}

/* console text of the reference, gstpeaq.c:1023-1035,1051-1060,1075 */
static void
print_movs (const GstPeaqAmd * self, const peaq_result * r)
{
  const gdouble *m = r->movs;
  if (!self->console_output)
    return;
  if (self->advanced)
    g_print ("RmsModDiffA = %f\nRmsNoiseLoudAsymA = %f\nSegmentalNMRB = %f\nEHSB = %f\nAvgLinDistA = %f\n",
        m[0], m[1], m[2], m[3], m[4]);
  else
    g_print ("   BandwidthRefB: %f\n  BandwidthTestB: %f\n      Total NMRB: %f\n"
        "    WinModDiff1B: %f\n            ADBB: %f\n            EHSB: %f\n"
        "    AvgModDiff1B: %f\n    AvgModDiff2B: %f\n   RmsNoiseLoudB: %f\n"
        "           MFPDB: %f\n  RelDistFramesB: %f\n",
        m[0], m[1], m[2], m[3], m[4], m[5], m[6], m[7], m[8], m[9], m[10]);
}

static gdouble
evaluate_odg (GstPeaqAmd * self)
{
  peaq_result r;
  read_results (self, &r);
  print_movs (self, &r);
  if (self->console_output)
    g_print ("Objective Difference Grade: %.3f\n", r.odg);
  return r.odg;
}

static void
gst_peaq_amd_get_property (GObject * obj, guint id, GValue * value, GParamSpec * pspec)
{
  GstPeaqAmd *self = GST_PEAQ_AMD (obj);
  peaq_result r;
  switch (id) {
    case PROP_PLAYBACK_LEVEL:
      g_value_set_double (value, self->playback_level);
      break;
    case PROP_ADVANCED:
      g_value_set_boolean (value, self->advanced);
      break;
    case PROP_CONSOLE_OUTPUT:
      g_value_set_boolean (value, self->console_output);
      break;
    case PROP_DI:
      read_results (self, &r);
      print_movs (self, &r);
      g_value_set_double (value, r.di);
      break;
    case PROP_ODG:
      g_value_set_double (value, evaluate_odg (self));
      break;
    case PROP_TOTALSNR:
      read_results (self, &r);
      g_value_set_double (value, r.totalsnr);
      break;
    default:
      G_OBJECT_WARN_INVALID_PROPERTY_ID (obj, id, pspec);
  }
}

static void
gst_peaq_amd_set_property (GObject * obj, guint id, const GValue * value, GParamSpec * pspec)
{
  GstPeaqAmd *self = GST_PEAQ_AMD (obj);
  switch (id) {
    case PROP_PLAYBACK_LEVEL:
      GST_OBJECT_LOCK (self);
      self->playback_level = g_value_get_double (value);
      /* the reference hands the level to both ear models at once (gstpeaq.c:508-515 ->
       * fftearmodel.c:305-314, fbearmodel.c:249-254): it applies from the next frame on */
      if (self->session) {
        if (peaq_session_set_level (self->session, self->playback_level) != PEAQ_OK)
          GST_WARNING_OBJECT (self, "libpeaq_amd: %s", peaq_last_error ());
      } else if (self->broker && self->playback_level != 92.) {
        /* the shared broker runs all its slots at one level: an element that wants another
         * one gets its own session -- possible as long as its slot has not seen any samples */
        if (!self->pushed)
          renew_session (self);
        else
          GST_WARNING_OBJECT (self, "playback_level changed mid-stream on a broker-hosted element: "
              "the new level applies from the next (re)negotiation");
      }
      GST_OBJECT_UNLOCK (self);
      post_pending_error (self);
      break;
    case PROP_ADVANCED:
      GST_OBJECT_LOCK (self);
      self->advanced = g_value_get_boolean (value);
      if (self->channels > 0)
        renew_session (self);
      GST_OBJECT_UNLOCK (self);
      post_pending_error (self);
      break;
    case PROP_CONSOLE_OUTPUT:
      self->console_output = g_value_get_boolean (value);
      break;
    default:
      G_OBJECT_WARN_INVALID_PROPERTY_ID (obj, id, pspec);
  }
}

static gint
pad_index (GstPeaqAmd * self, GstPad * pad)
{
  return pad == self->pads[0] ? 0 : 1;
}

static GstFlowReturn
gst_peaq_amd_chain (GstPad * pad, GstObject * parent, GstBuffer * buffer)
{
  GstPeaqAmd *self = GST_PEAQ_AMD (parent);
  GstMapInfo map;
  GstFlowReturn ret = GST_FLOW_OK;
  const gint idx = pad_index (self, pad);

  if (!gst_buffer_map (buffer, &map, GST_MAP_READ)) {
    gst_buffer_unref (buffer);
    return GST_FLOW_ERROR;
  }
  GST_OBJECT_LOCK (self);               /* the two streaming threads are serialised, gstpeaq.c:619,658 */
  self->eos[idx] = FALSE;
  self->pushed = TRUE;
  if (self->broker) {
    if (peaq_broker_push (self->broker, self->broker_sid, idx, (const float *) map.data,
            map.size / (sizeof (float) * self->channels)) != PEAQ_OK)
      ret = GST_FLOW_ERROR;
  } else if (!self->session) {
    ret = GST_FLOW_NOT_NEGOTIATED;
  } else if (peaq_session_push (self->session, idx, (const float *) map.data,
          map.size / (sizeof (float) * self->channels)) != PEAQ_OK) {
    ret = GST_FLOW_ERROR;
  }
  GST_OBJECT_UNLOCK (self);
  if (ret == GST_FLOW_ERROR && !self->failed) {
    self->failed = TRUE;
    GST_ELEMENT_ERROR (self, LIBRARY, FAILED, ("libpeaq_amd: %s", peaq_last_error ()), (NULL));
  }
  gst_buffer_unmap (buffer, &map);
  gst_buffer_unref (buffer);           /* the samples were copied by the engine */
  return ret;
}

static gboolean
gst_peaq_amd_set_caps (GstPeaqAmd * self, GstCaps * caps)
{
  gint channels = 0;
  gboolean ok = TRUE;
  if (!gst_structure_get_int (gst_caps_get_structure (caps, 0), "channels", &channels) || channels < 1
      || channels > 2) {
    GST_ELEMENT_ERROR (self, CORE, NEGOTIATION, ("peaq handles mono or stereo, got %d channels", channels), (NULL));
    return FALSE;
  }
  GST_OBJECT_LOCK (self);
  if (channels != self->channels || !(self->session || self->broker)) {
    self->channels = channels;
    ok = renew_session (self);
  }
  GST_OBJECT_UNLOCK (self);
  post_pending_error (self);
  return ok;
}

static gboolean
gst_peaq_amd_sink_event (GstPad * pad, GstObject * parent, GstEvent * event)
{
  GstPeaqAmd *self = GST_PEAQ_AMD (parent);
  const gint idx = pad_index (self, pad);
  gboolean ret;
  switch (GST_EVENT_TYPE (event)) {
    case GST_EVENT_EOS:{
      /* a sink posts EOS once ALL its pads are at EOS (gstpeaq.c:668-688).  Under the object lock: the two pads'
       * streaming threads get here at the same moment often enough (one process in thirty with 1024 elements), and
       * unlocked each could write its own flag, read the other's old value and leave the posting to the other --
       * a pipeline that never ends. */
      gboolean both;
      GST_OBJECT_LOCK (self);
      self->eos[idx] = TRUE;
      both = self->eos[0] && self->eos[1];
      GST_OBJECT_UNLOCK (self);
      ret = TRUE;
      if (both) {
        GstMessage *msg = gst_message_new_eos (parent);
        gst_message_set_seqnum (msg, gst_event_get_seqnum (event));
        ret = gst_element_post_message (GST_ELEMENT (self), msg);
      }
      gst_event_unref (event);
      return ret;
    }
    case GST_EVENT_CAPS:{
      GstCaps *caps;
      gst_event_parse_caps (event, &caps);
      /* both inputs must carry the same number of channels: only accept what the
       * other pad's upstream can also deliver (gstpeaq.c:689-708) */
      ret = gst_pad_peer_query_accept_caps (self->pads[1 - idx], caps) && gst_peaq_amd_set_caps (self, caps);
      gst_event_unref (event);
      return ret;
    }
    default:
      return gst_pad_event_default (pad, parent, event);
  }
}

static gboolean
gst_peaq_amd_sink_query (GstPad * pad, GstObject * parent, GstQuery * query)
{
  GstPeaqAmd *self = GST_PEAQ_AMD (parent);
  if (GST_QUERY_TYPE (query) == GST_QUERY_CAPS) {
    /* offer the template caps restricted to what the OTHER input can produce */
    const gint idx = pad_index (self, pad);
    GstCaps *filter, *tmpl, *other, *result;
    gst_query_parse_caps (query, &filter);
    tmpl = gst_pad_get_pad_template_caps (pad);
    other = gst_pad_peer_query_caps (self->pads[1 - idx], filter);
    result = gst_caps_intersect (tmpl, other);
    gst_caps_unref (tmpl);
    gst_caps_unref (other);
    gst_query_set_caps_result (query, result);
    gst_caps_unref (result);
    return TRUE;
  }
  return gst_pad_query_default (pad, parent, query);
}

static GstStateChangeReturn
gst_peaq_amd_change_state (GstElement * element, GstStateChange transition)
{
  GstPeaqAmd *self = GST_PEAQ_AMD (element);
  if (transition == GST_STATE_CHANGE_PAUSED_TO_READY) {
    /* do_flush + calculate_odg, gstpeaq.c:764-778 */
    if ((self->broker && peaq_broker_flush (self->broker, self->broker_sid) != PEAQ_OK)
        || (self->session && peaq_session_flush (self->session) != PEAQ_OK))
      GST_ERROR_OBJECT (self, "libpeaq_amd: %s", peaq_last_error ());
    evaluate_odg (self);
  }
  return GST_ELEMENT_CLASS (gst_peaq_amd_parent_class)->change_state (element, transition);
}

static void
gst_peaq_amd_finalize (GObject * obj)
{
  GstPeaqAmd *self = GST_PEAQ_AMD (obj);
  drop_session (self);
  g_free (self->pending_error);
  G_OBJECT_CLASS (gst_peaq_amd_parent_class)->finalize (obj);
}

static void
gst_peaq_amd_class_init (GstPeaqAmdClass * klass)
{
  GObjectClass *oc = G_OBJECT_CLASS (klass);
  GstElementClass *ec = GST_ELEMENT_CLASS (klass);

  oc->get_property = gst_peaq_amd_get_property;
  oc->set_property = gst_peaq_amd_set_property;
  oc->finalize = gst_peaq_amd_finalize;
  ec->change_state = gst_peaq_amd_change_state;

  /* names, ranges and defaults as in gstpeaq.c:273-317 ("playback_level" with an underscore) */
  g_object_class_install_property (oc, PROP_PLAYBACK_LEVEL,
      g_param_spec_double ("playback_level", "playback level", "Playback level in dB", 0, 130, 92,
          G_PARAM_READWRITE | G_PARAM_CONSTRUCT));
  g_object_class_install_property (oc, PROP_ADVANCED,
      g_param_spec_boolean ("advanced", "Advanced mode enabled", "True if advanced mode is used", FALSE,
          G_PARAM_READWRITE | G_PARAM_CONSTRUCT));
  g_object_class_install_property (oc, PROP_DI,
      g_param_spec_double ("di", "distortion index", "Distortion Index", -G_MAXDOUBLE, G_MAXDOUBLE, 0,
          G_PARAM_READABLE));
  g_object_class_install_property (oc, PROP_ODG,
      g_param_spec_double ("odg", "objective difference grade", "Objective Difference Grade", -G_MAXDOUBLE,
          G_MAXDOUBLE, 0, G_PARAM_READABLE));
  g_object_class_install_property (oc, PROP_TOTALSNR,
      g_param_spec_double ("totalsnr", "the overall SNR in dB", "the overall signal to noise ratio in dB",
          -G_MAXDOUBLE, G_MAXDOUBLE, 0, G_PARAM_READABLE));
  g_object_class_install_property (oc, PROP_CONSOLE_OUTPUT,
      g_param_spec_boolean ("console-output", "console output", "Enable or disable console output", TRUE,
          G_PARAM_READWRITE | G_PARAM_CONSTRUCT));

  gst_element_class_add_static_pad_template (ec, &ref_template);
  gst_element_class_add_static_pad_template (ec, &test_template);
  gst_element_class_set_static_metadata (ec, "Perceptual evaluation of audio quality (MI355X)", "Sink/Audio",
      "Compute objective audio quality measures (ITU-R BS.1387) on an AMD GPU", "gstpeaq_amd");
}

static void
gst_peaq_amd_init (GstPeaqAmd * self)
{
  static GstStaticPadTemplate *tmpl[2] = { &ref_template, &test_template };
  gint i;
  for (i = 0; i < 2; i++) {
    self->pads[i] = gst_pad_new_from_static_template (tmpl[i], i ? "test" : "ref");
    gst_pad_set_chain_function (self->pads[i], gst_peaq_amd_chain);
    gst_pad_set_event_function (self->pads[i], gst_peaq_amd_sink_event);
    gst_pad_set_query_function (self->pads[i], gst_peaq_amd_sink_query);
    gst_element_add_pad (GST_ELEMENT (self), self->pads[i]);
  }
  GST_OBJECT_FLAG_SET (self, GST_ELEMENT_FLAG_SINK);
  self->channels = 0;
  self->session = NULL;
  self->broker = NULL;
  self->broker_sid = -1;
  self->failed = FALSE;
  self->pushed = FALSE;
  self->pending_error = NULL;
}

static gboolean
plugin_init (GstPlugin * plugin)
{
  GST_DEBUG_CATEGORY_INIT (peaq_amd_debug, "peaq", 0, "PEAQ on MI355X");
  return gst_element_register (plugin, "peaq", GST_RANK_NONE, gst_peaq_amd_get_type ());
}

GST_PLUGIN_DEFINE (GST_VERSION_MAJOR, GST_VERSION_MINOR, peaq,
    "Perceptual evaluation of audio quality (ITU-R BS.1387) on AMD MI355X",
    plugin_init, PACKAGE_VERSION, "LGPL", PACKAGE, "https://github.com/HSU-ANT/gstpeaq")
