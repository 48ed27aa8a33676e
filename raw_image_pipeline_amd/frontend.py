"""Streaming front-end: the counterpart of the reference's ROS node (raw_image_pipeline_ros/src/
raw_image_pipeline_ros.cpp) without ROS -- per-camera parameter set with the node's names and defaults,
the image callback with its three output taps (``<type>_rect``, ``debayered``, ``<type>``) plus
CameraInfo, slow-topic decimation, the ``reset_white_balance`` control, output encoding BGR/RGB/
passthrough -- and a multi-camera rig that shards cameras over the GPUs of a node (camera c -> GPU
c mod N).  Transport (subscribing / publishing) is the caller's business: the callback
returns the messages it would have published.
"""
from collections import OrderedDict

import numpy as np

from .pipeline import OutputPool, RawImagePipeline

# readParameter defaults of RawImagePipelineRos::loadParams (raw_image_pipeline_ros.cpp:36-182)
NODE_DEFAULTS = OrderedDict([
    ("input_type", "color"), ("output_prefix", "/camera"), ("transport", "raw"),
    ("output_encoding", "BGR"), ("output_frame", "passthrough"), ("skip_number_of_images_for_slow_topic", -1),
    ("use_gpu", True), ("debug", False),
    ("debayer/enabled", True), ("debayer/encoding", "auto"),
    ("flip/enabled", False), ("flip/angle", 0),
    ("white_balance/enabled", False), ("white_balance/method", "simple"), ("white_balance/clipping_percentile", 10.0),
    ("white_balance/saturation_bright_thr", 0.9), ("white_balance/saturation_dark_thr", 0.1),
    ("white_balance/temporal_consistency", False),
    ("color_calibration/enabled", False), ("color_calibration/calibration_file", ""),
    ("color_calibration/calibration_matrix/data", [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]),
    ("color_calibration/calibration_bias/data", [0.0, 0.0, 0.0]),
    ("gamma_correction/enabled", False), ("gamma_correction/method", "default"), ("gamma_correction/k", 0.8),
    ("vignetting_correction/enabled", False), ("vignetting_correction/scale", 1.0), ("vignetting_correction/a2", 1.0),
    ("vignetting_correction/a4", 1.0),
    ("color_enhancer/enabled", False), ("color_enhancer/hue_gain", 1.0), ("color_enhancer/saturation_gain", 1.0),
    ("color_enhancer/value_gain", 1.0),
    ("undistortion/enabled", False), ("undistortion/balance", 0.0), ("undistortion/fov_scale", 1.0),
    ("undistortion/calibration_file", ""), ("undistortion/image_width", 640), ("undistortion/image_height", 480),
    ("undistortion/camera_matrix/data", [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]),
    ("undistortion/distortion_coefficients/data", [0.0, 0.0, 0.0, 0.0]), ("undistortion/distortion_model", "none"),
    ("undistortion/rectification_matrix/data", [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]),
    ("undistortion/projection_matrix/data", [1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0]),
])


def transport_hint_from_topic(topic):
    """getTransportHintFromTopic (:349-362): ".../image/compressed" -> (".../image", "compressed")."""
    ind = topic.find("compressed")
    if ind != -1:
        return topic[:ind - 1], topic[ind:]
    return topic, "raw"


class CameraStream:
    """One camera = one pipeline handle = one HIP stream (RawImagePipelineRos, one node per camera)."""

    def __init__(self, params=None, device=0, ccc_model=None, pipeline=None):
        unknown = set(params or ()) - set(NODE_DEFAULTS)
        if unknown:
            raise KeyError("unknown parameter(s): %s" % sorted(unknown))
        self.params = OrderedDict(NODE_DEFAULTS)
        self.params.update(params or {})
        p = self.params
        self.input_type = p["input_type"]
        self.output_prefix = p["output_prefix"]
        self.transport = p["transport"]
        self.output_encoding = p["output_encoding"]
        self.output_frame = p["output_frame"]
        self.skip = int(p["skip_number_of_images_for_slow_topic"])
        self._skipped = 0       # skipped_images_for_slow_topic_
        self._skipped_rect = 0  # skipped_images_for_slow_topic_rect_
        # raw_image_pipeline_ = std::make_unique<RawImagePipeline>(use_gpu)  (:52): the one-argument constructor
        self.pipe = pipeline if pipeline is not None else RawImagePipeline(bool(p["use_gpu"]), device=device)
        # The published images are deep copies (the reference clones every one, raw_image_pipeline.cpp:222-236).  Their arrays
        # are recycled once the subscriber has dropped them -- a fresh 15 MB array costs more than filling it -- and they are
        # page-locked: the synchronous path downloads into them at the full PCIe rate, and on the pipelined path the downloads
        # are written into them directly (rip_submit_to): no pinned buffer of the handle in between, no memcpy at collect()
        if self.pipe.out_pool is None:
            self.pipe.out_pool = OutputPool(limit=16, pinned=True)
        self._pinned_pool = self.pipe.out_pool if self.pipe.out_pool.pinned else None
        self._configure()
        if ccc_model is not None:
            self.pipe.set_ccc_model(*ccc_model)

    def _configure(self):
        p, pipe = self.params, self.pipe
        pipe.set_debug(p["debug"])
        pipe.set_debayer(p["debayer/enabled"])
        pipe.set_debayer_encoding(p["debayer/encoding"])
        pipe.set_flip(p["flip/enabled"])
        pipe.set_flip_angle(p["flip/angle"])
        pipe.set_white_balance(p["white_balance/enabled"])
        pipe.set_white_balance_method(p["white_balance/method"])
        pipe.set_white_balance_percentile(p["white_balance/clipping_percentile"])
        pipe.set_white_balance_saturation_threshold(p["white_balance/saturation_bright_thr"], p["white_balance/saturation_dark_thr"])
        pipe.set_white_balance_temporal_consistency(p["white_balance/temporal_consistency"])
        pipe.set_color_calibration(p["color_calibration/enabled"])
        if p["color_calibration/enabled"]:
            pipe.load_color_calibration(p["color_calibration/calibration_file"])
            if not p["color_calibration/calibration_file"]:
                pipe.set_color_calibration_matrix(p["color_calibration/calibration_matrix/data"])
                pipe.set_color_calibration_bias(p["color_calibration/calibration_bias/data"])
        pipe.set_gamma_correction(p["gamma_correction/enabled"])
        pipe.set_gamma_correction_method(p["gamma_correction/method"])
        pipe.set_gamma_correction_k(p["gamma_correction/k"])
        pipe.set_vignetting_correction(p["vignetting_correction/enabled"])
        pipe.set_vignetting_correction_parameters(p["vignetting_correction/scale"], p["vignetting_correction/a2"],
                                                  p["vignetting_correction/a4"])
        pipe.set_color_enhancer(p["color_enhancer/enabled"])
        pipe.set_color_enhancer_hue_gain(p["color_enhancer/hue_gain"])
        pipe.set_color_enhancer_saturation_gain(p["color_enhancer/saturation_gain"])
        pipe.set_color_enhancer_value_gain(p["color_enhancer/value_gain"])
        pipe.set_undistortion(p["undistortion/enabled"])
        pipe.set_undistortion_balance(p["undistortion/balance"])
        pipe.set_undistortion_fov_scale(p["undistortion/fov_scale"])
        pipe.load_camera_calibration(p["undistortion/calibration_file"])
        if not p["undistortion/calibration_file"]:
            # the node reads image_height from the image_width parameter (:156-157); not reproduced
            pipe.set_undistortion_image_size(p["undistortion/image_width"], p["undistortion/image_height"])
            pipe.set_undistortion_camera_matrix(p["undistortion/camera_matrix/data"])
            pipe.set_undistortion_distortion_coeffs(p["undistortion/distortion_coefficients/data"])
            pipe.set_undistortion_distortion_model(p["undistortion/distortion_model"])
            pipe.set_undistortion_rectification_matrix(p["undistortion/rectification_matrix/data"])
            pipe.set_undistortion_projection_matrix(p["undistortion/projection_matrix/data"])
            pipe.init_undistortion()

    # ---- topics (setupRos :184-217) ------------------------------------------------------------------
    def topics(self):
        t = []
        if self.pipe.is_undistortion_enabled():
            t += [self.output_prefix + "/" + self.input_type + "_rect/image", self.output_prefix + "/" + self.input_type + "_rect/image/slow"]
        if self.input_type == "color":
            t += [self.output_prefix + "/debayered/image", self.output_prefix + "/debayered/slow"]
        t += [self.output_prefix + "/" + self.input_type + "/image", self.output_prefix + "/" + self.input_type + "/image/slow"]
        return t

    def reset_white_balance(self):
        """~reset_white_balance service (:290-295)."""
        self.pipe.reset_white_balance_temporal_consistency()
        return True, "White balance resetted"

    # ---- the image callback (:219-288) -------------------------------------------------------------------
    def _sync_taps(self):
        """Keeps only the taps the node publishes (:245-287): the debayered image for colour inputs with the debayer on, the
        pre-undistortion colour image when undistortion is on, the final image always.  (The reference clones all three per
        frame whatever is published; here a tap that nobody reads costs device writes and, on the pipelined path, a download.)"""
        from .pipeline import TAP_COLOR, TAP_DEBAYERED, TAP_PROCESSED
        mask = TAP_PROCESSED
        if self.input_type == "color" and self.pipe.is_debayer_enabled():
            mask |= TAP_DEBAYERED
        if self.pipe.is_undistortion_enabled():
            mask |= TAP_COLOR
        if mask != getattr(self, "_tap_mask", None):
            self.pipe.set_taps(mask)
            self.pipe.set_tap_download(mask)  # every kept tap is published: let it travel with the result of a submit()
            self._tap_mask = mask

    def on_image(self, image, encoding, stamp=0.0, frame_id="camera"):
        """Processes one frame and returns the messages the node would publish, in publishing order:
        a list of dicts {topic, image, encoding, camera_info | None}."""
        img = np.asarray(image)
        if img.size == 0:
            return []  # ROS_WARN("image empty")
        if self.transport != "raw":
            encoding = "bgr8"  # cv_bridge::toCvCopy(image_msg, "bgr8") for compressed transports
        self._sync_taps()
        processed = self.pipe.apply(img.copy(), encoding)
        return self._messages(processed, self.pipe.last_encoding, stamp, frame_id)

    def submit(self, image, encoding, stamp=0.0, frame_id="camera"):
        """First half of the callback: enqueues upload, kernels and the download of the result AND of the taps the node
        publishes (rip_submit) and returns at once.  The frame is read before the call returns.  False for an empty image."""
        img = np.ascontiguousarray(image)
        if img.size == 0:
            return False
        if self.transport != "raw":
            encoding = "bgr8"
        self._sync_taps()
        dst = {}
        if self._pinned_pool is not None:
            from .pipeline import TAP_COLOR, TAP_DEBAYERED
            rows, cols = img.shape[:2]
            cn = 1 if img.ndim == 2 else img.shape[2]
            r, c, k, enc_out = self.pipe.query_output(rows, cols, cn, encoding)
            if not enc_out.endswith("16"):
                dst["out"] = self._pinned_pool.take((r, c) if k == 1 else (r, c, k), or_none=True)
                tr, tc, tk = self.pipe.query_taps(rows, cols, cn, encoding)
                if self._tap_mask & TAP_DEBAYERED:
                    dst["tap_debayered"] = self._pinned_pool.take((tr, tc) if tk == 1 else (tr, tc, tk), or_none=True)
                if self._tap_mask & TAP_COLOR:
                    dst["tap_color"] = self._pinned_pool.take((tr, tc) if tk == 1 else (tr, tc, tk), or_none=True)
        ticket = self.pipe.submit(img, encoding, **{key: arr for key, arr in dst.items() if arr is not None})
        self._inflight = getattr(self, "_inflight", [])
        self._inflight.append((ticket, stamp, frame_id))
        return True

    def collect(self, copy=True):
        """Second half: waits for the OLDEST frame in flight and returns its messages ([] when nothing is in flight).  With
        ``copy=False`` the images are read-only views of the handle's pinned buffers -- what a publisher that serialises the
        message at once wants -- valid until the next collect() of this camera."""
        inflight = getattr(self, "_inflight", [])
        if not inflight:
            return []
        ticket, stamp, frame_id = inflight.pop(0)
        processed = self.pipe.collect(ticket, copy=copy)
        return self._messages(processed, self.pipe.last_encoding, stamp, frame_id, copy=copy)

    def on_image_pipelined(self, image, encoding, stamp=0.0, frame_id="camera", copy=True):
        """The same callback with one frame kept in flight (rip_submit / rip_collect): uploads, kernels and downloads of
        neighbouring frames overlap, and the call returns the messages of the PREVIOUS frame (an empty list for the first
        one; flush() delivers the last).  Frames are processed in arrival order, so the white-balance filter sees the
        same sequence as with on_image()."""
        had = len(getattr(self, "_inflight", []))
        if not self.submit(image, encoding, stamp, frame_id):
            return []
        return self.collect(copy=copy) if had else []

    def flush(self, copy=True):
        """Messages of the frame still in flight, if any."""
        return self.collect(copy=copy)

    def _messages(self, processed, enc, stamp, frame_id, copy=True):
        out = []
        pipe = self.pipe
        if pipe.is_undistortion_enabled():
            self._publish(out, processed, enc, stamp, frame_id, self.input_type + "_rect/image", self.input_type + "_rect/image/slow",
                          pipe.get_rect_image_height(), pipe.get_rect_image_width(), pipe.get_rect_distortion_model(),
                          pipe.get_rect_distortion_coefficients(), pipe.get_rect_camera_matrix(),
                          pipe.get_rect_rectification_matrix(), pipe.get_rect_projection_matrix(), "_skipped_rect")
        dist = (pipe.get_dist_image_height(), pipe.get_dist_image_width(), pipe.get_dist_distortion_model(),
                pipe.get_dist_distortion_coefficients(), pipe.get_dist_camera_matrix(), pipe.get_dist_rectification_matrix(),
                pipe.get_dist_projection_matrix())
        if self.input_type == "color" and pipe.is_debayer_enabled():
            self._publish(out, pipe.get_dist_debayered_image(copy=copy), enc, stamp, frame_id, "debayered/image", "debayered/slow", *dist,
                          "_skipped")
        # getProcessedImage() (:283) is the image apply() has just returned: no second read of it
        color = pipe.get_dist_color_image(copy=copy) if pipe.is_undistortion_enabled() else processed
        self._publish(out, color, enc, stamp, frame_id, self.input_type + "/image", self.input_type + "/image/slow", *dist, "_skipped")
        return out

    def _publish(self, out, image, encoding, stamp, frame_id, topic, slow_topic, height, width, model, D, K, R, P, counter):
        """publishColorImage (:297-347)."""
        if self.output_encoding == "RGB" and image.ndim == 3:
            image = image[..., ::-1].copy()  # cv::cvtColor(BGR2RGB)
        if image.ndim == 3 and image.shape[2] == 3:
            if self.output_encoding == "RGB":
                encoding = "rgb8"
            elif self.output_encoding == "BGR":
                encoding = "bgr8"
            elif self.output_encoding != "passthrough":
                raise ValueError("Found invalid image encoding: %s, make sure to set a supported ouput encoding "
                                 "('RGB', 'BGR', or 'passthrough')" % self.output_encoding)
        fid = frame_id if self.output_frame == "passthrough" else self.output_frame
        info = {"header": {"stamp": stamp, "frame_id": fid}, "height": int(height), "width": int(width),
                "distortion_model": model, "D": [float(v) for v in np.asarray(D).ravel()],
                "K": [float(v) for v in np.asarray(K).ravel()], "R": [float(v) for v in np.asarray(R).ravel()],
                "P": [float(v) for v in np.asarray(P).ravel()]}
        msg = {"topic": self.output_prefix + "/" + topic, "image": image, "encoding": encoding, "camera_info": info,
               "header": {"stamp": stamp, "frame_id": fid}}
        out.append(msg)
        skipped = getattr(self, counter)
        if skipped >= self.skip or self.skip <= 0:
            out.append({"topic": self.output_prefix + "/" + slow_topic, "image": image, "encoding": encoding, "camera_info": None,
                        "header": {"stamp": stamp, "frame_id": fid}})
            setattr(self, counter, 0)
        else:
            setattr(self, counter, skipped + 1)


class CameraRig:
    """N cameras sharded over the visible GPUs: camera c is owned by device c mod n_devices, keeps its own
    handle (maps, LUTs, Kalman state resident on that device) and its own HIP stream."""

    def __init__(self, camera_params, n_devices=None, ccc_model=None):
        import torch
        n_devices = n_devices or max(1, torch.cuda.device_count())
        self.streams = []
        self.devices = []
        self.hip_streams = []
        self._pool = None
        from .sharding import streams_of_rank
        # camera c is owned by device c mod n_devices -- the same rule a multi-process job applies per rank (sharding.py)
        owner = {c: d for d in range(n_devices) for c in streams_of_rank(len(camera_params), n_devices, d)}
        for c, params in enumerate(camera_params):
            dev = owner[c]
            with torch.cuda.device(dev):
                s = torch.cuda.Stream(device=dev)
            cam = CameraStream(params, device=dev, ccc_model=ccc_model)
            cam.pipe.set_stream(s)
            self.streams.append(cam)
            self.devices.append(dev)
            self.hip_streams.append(s)

    def on_images(self, images, encodings, stamp=0.0, parallel=True, mode=None, copy=True):
        """One frame per camera (the synchronised trigger of a multi-camera rig).  Modes:

        * ``"pipelined"`` (default): ONE thread submits every camera's frame (rip_submit: upload, kernels and the downloads
          of result + published taps are only enqueued, each camera on its own handle and streams) and then collects them
          in camera order, so camera c's downloads overlap camera c + 1's upload and kernels -- the host path is PCIe-bound
           and the link is full duplex.  No thread, no GIL hand-over, no per-call pool latency.
        * ``"threaded"``: one thread per camera around the synchronous callback (the shape of one ROS node per camera);
          kept as an option -- on a many-core host its wake-ups cost more than the overlap gives at small frames.
        * ``"sequential"`` (or ``parallel=False``): camera after camera, synchronously.

        ``copy=False`` (pipelined only): images are read-only views of pinned memory, valid until the next trigger."""
        if mode is None:
            mode = "pipelined" if parallel else "sequential"
        jobs = list(enumerate(zip(self.streams, images, encodings)))
        run = lambda job: job[1][0].on_image(job[1][1], job[1][2], stamp=stamp, frame_id="cam%d" % job[0])
        if mode == "sequential" or len(jobs) < 2:
            return [run(j) for j in jobs]
        if mode == "pipelined":
            sent = []
            try:
                for c, (cam, img, enc) in jobs:
                    sent.append(cam.submit(img, enc, stamp=stamp, frame_id="cam%d" % c))
            except BaseException:
                # camera k refused its frame (bad encoding, ring full): cameras 0 .. k-1 already have this trigger's frame in
                # flight.  Collect them before the error leaves -- left in flight they would come back on the NEXT trigger and
                # those cameras would publish the previous trigger's images from then on (camera_rig.hpp does the same)
                for ok, (c, (cam, img, enc)) in zip(sent, jobs):
                    if ok:
                        try:
                            cam.collect(copy=False)
                        except Exception:  # noqa: BLE001 -- the first error is the one to report
                            pass
                raise
            return [cam.collect(copy=copy) if ok else [] for ok, (c, (cam, img, enc)) in zip(sent, jobs)]
        if mode != "threaded":
            raise ValueError("mode must be 'pipelined', 'threaded' or 'sequential'")
        if self._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=len(self.streams), thread_name_prefix="rip-camera")
        return list(self._pool.map(run, jobs))

    def process_resident(self, batches, encodings, outs=None):
        """Device-resident batches, one per camera, each on its owner's device and stream; asynchronous."""
        res = []
        for c, (cam, frames, enc) in enumerate(zip(self.streams, batches, encodings)):
            res.append(cam.pipe.apply_device(frames, enc, out=None if outs is None else outs[c]))
        return res

    def synchronize(self):
        for s in self.hip_streams:
            s.synchronize()
