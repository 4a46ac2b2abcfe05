// IVideoCodec over libjsplayer_amd.so: the class Manager.video_info_cb constructs instead of MSVideo1_16bit /
// MSVideo1_8bit / ScreenPressor (Manager.hx:105-110) on the hxcpp target:
//
//     pool = new FramePool(vi.X, vi.Y, num_buffers + 1);              // instead of new Int32Array(...) x9, Manager.hx:114-118
//     buffers = pool.buffers;
//     switch (vi.codec) {
//         case codec_screenpressor: decoder = new NativeCodec(NativeCodec.SCREENPRESSOR, vi.X, vi.Y, vi.bpp, null, pool);
//         case codec_msvc16:        decoder = new NativeCodec(NativeCodec.MSVIDEO1_16, vi.X, vi.Y, 16, null, pool);
//         case codec_msvc8:         decoder = new NativeCodec(NativeCodec.MSVIDEO1_8, vi.X, vi.Y, 8, vi.palette, pool);
//     }
//     decoder.Preinit(INSIGNIFICANT_LINES);                            // Manager.hx:128, unchanged
//
// Everything else in Manager.worker (Manager.hx:454-539) stays as it is: the nine methods below have the reference's
// names, arguments and results.  Compressed frames arrive as haxe.io.Bytes (the hxcpp stand-in for js.lib.Uint8Array in
// IVideoCodec.hx:21,24,26).
#if cpp
package;

import cpp.RawConstPointer;
import cpp.RawPointer;
import cpp.UInt8;
import haxe.io.Bytes;
import IVideoCodec;   // DecoderState, PFrameResult (IVideoCodec.hx:5-14)

class NativeCodec implements IVideoCodec {
    public static inline var MSVIDEO1_16 = 1;      // JSP_CODEC_MSVIDEO1_16
    public static inline var MSVIDEO1_8 = 2;       // JSP_CODEC_MSVIDEO1_8
    public static inline var SCREENPRESSOR = 3;    // JSP_CODEC_SCREENPRESSOR

    var h:RawPointer<JspCodec>;
    var pool:FramePool;

    public function new(kind:Int, width:Int, height:Int, bpp:Int, palette:Null<Bytes>, pool:FramePool, device:Int = 0) {
        this.pool = pool;
        var pal:RawConstPointer<UInt8> = palette != null ? bytesPtr(palette) : null;
        h = JspNative.create(kind, width, height, bpp, pal, palette != null ? palette.length : 0, device);
        if (h == null) throw "jsp_codec_create: " + JspNative.lastError().toString();
        if (kind != SCREENPRESSOR) JspNative.setOption(h, "msv1_parse", "gpu");   // descriptor-free on-GPU parse
    }

    static inline function bytesPtr(b:Bytes):RawConstPointer<UInt8> {
        return cast cpp.NativeArray.address(b.getData(), 0).constRaw;
    }

    static inline function stateOf(rc:Int):DecoderState {
        return switch (rc) { case 0: zero_state; case 1: in_progress; default: error_occured; }
    }

    // ---- IVideoCodec (IVideoCodec.hx:16-29) ---------------------------------------------------------------------------
    public function Preinit(insignificant_lines:Int):Void {
        JspNative.preinit(h, insignificant_lines);
        JspNative.setOption(h, "key_frame_compare", Std.string(insignificant_lines));   // key frames compared while they decode (KeyFrameDiffers)
    }

    /** Manager.frames_differ_significantly's pixel loop (Manager.hx:413-419) comes with the decode: after Preinit every key frame is
        compared with the frame before it from `insignificant_lines` on (option "key_frame_compare"); null = nothing to compare with
        (the Manager's `prev == null` / first-frame cases). */
    public function KeyFrameDiffers():Null<Bool> {
        var v = JspNative.keyFrameDiffers(h);
        return v < 0 ? null : v != 0;
    }

    public function PreviousFrame():FrameBuffer {
        return pool.find(JspNative.previousFrame(h));
    }

    public function IsKeyFrame(data:Bytes):Bool {
        return JspNative.isKeyFrame(h, bytesPtr(data), data.length) != 0;
    }

    public function State():DecoderState {
        return stateOf(JspNative.state(h));
    }

    public function ContinueI():DecoderState {
        return stateOf(JspNative.continueI(h));
    }

    public function DecompressI(src:Bytes, dst:FrameBuffer):DecoderState {
        return stateOf(JspNative.decompressI(h, bytesPtr(src), src.length, dst.ptr));
    }

    public function DecompressP(src:Bytes, dst:FrameBuffer):PFrameResult {
        var dataPnt:RawPointer<cpp.Int32> = null;
        var signif:Int = 0;
        var rc = JspNative.decompressP(h, bytesPtr(src), src.length, dst.ptr, cpp.RawPointer.addressOf(dataPnt), cpp.RawPointer.addressOf(signif));
        // the one place the reference raises out of DecompressP (a skip code before any frame exists, MSVideo1.hx:79):
        if (rc != 0) throw "DecompressP: " + JspNative.lastError().toString();
        return { data_pnt: pool.find(dataPnt), significant_changes: signif != 0 };
    }

    public function NeedsIndex():Bool {
        return JspNative.needsIndex(h) != 0;
    }

    public function StopAndClean():Void {
        if (h != null) JspNative.destroy(h);
        h = null;
    }

    // ---- optional: decode ahead of display (jsp_decompress_*_async / jsp_wait) ------------------------------------------
    /** Queue a frame; `src` and `dst` must stay untouched until wait(ticket).  Returns the ticket. */
    public function Submit(src:Bytes, dst:FrameBuffer, key:Bool):haxe.Int64 {
        var ticket:cpp.UInt64 = 0;
        var rc = key ? JspNative.decompressIAsync(h, bytesPtr(src), src.length, dst.ptr, cpp.RawPointer.addressOf(ticket))
                     : JspNative.decompressPAsync(h, bytesPtr(src), src.length, dst.ptr, cpp.RawPointer.addressOf(ticket));
        if (rc != 0) throw "submit: " + JspNative.lastError().toString();
        return cast ticket;
    }

    /** What DecompressI / DecompressP would have returned for the frame queued under `ticket` (tickets in order). */
    public function Wait(ticket:haxe.Int64):{state:DecoderState, result:PFrameResult} {
        var dataPnt:RawPointer<cpp.Int32> = null;
        var signif:Int = 0;
        var rc = JspNative.wait(h, cast ticket, cpp.RawPointer.addressOf(dataPnt), cpp.RawPointer.addressOf(signif));
        return { state: stateOf(rc), result: { data_pnt: pool.find(dataPnt), significant_changes: signif != 0 } };
    }
}
#end
