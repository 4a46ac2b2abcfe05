// hxcpp externs over include/jsplayer_amd.h — one row per export the three classes below use.
// Build: add  <files id="haxe"> <compilerflag value="-I${JSPLAYER_AMD}/include"/> </files>
//             <target id="haxe"> <lib name="-L${JSPLAYER_AMD}/jsplayer_amd"/> <lib name="-ljsplayer_amd"/> </target>
// to the project's Build.xml (or keep the @:buildXml below and set JSPLAYER_AMD in the environment).
#if cpp
package;

import cpp.ConstCharStar;
import cpp.RawConstPointer;
import cpp.RawPointer;
import cpp.SizeT;
import cpp.UInt64;
import cpp.UInt8;

@:keep
@:include("jsplayer_amd.h")
@:buildXml('
<files id="haxe"><compilerflag value="-I${JSPLAYER_AMD}/include"/></files>
<target id="haxe"><lib name="-L${JSPLAYER_AMD}/jsplayer_amd"/><lib name="-ljsplayer_amd"/></target>
')
extern class JspNative {
    // IVideoCodec.hx:16-29 <-> C ABI, see INTEGRATION.md §1
    @:native("jsp_codec_create")       static function create(kind:Int, w:Int, h:Int, bpp:Int, palette:RawConstPointer<UInt8>, paletteBytes:Int, device:Int):RawPointer<JspCodec>;
    @:native("jsp_codec_destroy")      static function destroy(c:RawPointer<JspCodec>):Void;
    @:native("jsp_preinit")            static function preinit(c:RawPointer<JspCodec>, lines:Int):Int;
    @:native("jsp_previous_frame")     static function previousFrame(c:RawPointer<JspCodec>):RawPointer<cpp.Int32>;
    @:native("jsp_is_key_frame")       static function isKeyFrame(c:RawPointer<JspCodec>, src:RawConstPointer<UInt8>, n:SizeT):Int;
    @:native("jsp_state")              static function state(c:RawPointer<JspCodec>):Int;
    @:native("jsp_continue_i")         static function continueI(c:RawPointer<JspCodec>):Int;
    @:native("jsp_decompress_i")       static function decompressI(c:RawPointer<JspCodec>, src:RawConstPointer<UInt8>, n:SizeT, dst:RawPointer<cpp.Int32>):Int;
    @:native("jsp_decompress_p")       static function decompressP(c:RawPointer<JspCodec>, src:RawConstPointer<UInt8>, n:SizeT, dst:RawPointer<cpp.Int32>,
                                                                   dataPnt:RawPointer<RawPointer<cpp.Int32>>, significant:RawPointer<Int>):Int;
    @:native("jsp_needs_index")        static function needsIndex(c:RawPointer<JspCodec>):Int;
    @:native("jsp_last_error")         static function lastError():ConstCharStar;
    @:native("jsp_set_option")         static function setOption(c:RawPointer<JspCodec>, key:ConstCharStar, value:ConstCharStar):Int;
    @:native("jsp_counter")            static function counter(c:RawPointer<JspCodec>, name:ConstCharStar):cpp.Int64;   // diagnostics: "async_reruns", "lookback_fallbacks"
    // the asynchronous form of DecompressI / DecompressP (optional: a Manager that decodes ahead of display)
    @:native("jsp_decompress_i_async") static function decompressIAsync(c:RawPointer<JspCodec>, src:RawConstPointer<UInt8>, n:SizeT, dst:RawPointer<cpp.Int32>, ticket:RawPointer<UInt64>):Int;
    @:native("jsp_decompress_p_async") static function decompressPAsync(c:RawPointer<JspCodec>, src:RawConstPointer<UInt8>, n:SizeT, dst:RawPointer<cpp.Int32>, ticket:RawPointer<UInt64>):Int;
    @:native("jsp_prefetch")           static function prefetch(c:RawPointer<JspCodec>, host:RawConstPointer<UInt8>, bytes:SizeT):Int;   // a stretch of the file ahead of the frames submitted next: one copy instead of one per frame
    @:native("jsp_wait")               static function wait(c:RawPointer<JspCodec>, ticket:UInt64, dataPnt:RawPointer<RawPointer<cpp.Int32>>, significant:RawPointer<Int>):Int;
    // frame pool in HBM (Manager.hx:114-118) and the two Manager passes that follow the codec
    @:native("jsp_key_frame_differs")  static function keyFrameDiffers(c:RawPointer<JspCodec>):Int;
    @:native("jsp_device_count")       static function deviceCount():Int;
    @:native("jsp_assign_stream")      static function assignStream(streamIndex:Int, devices:RawPointer<Int>, ndev:Int):Int;
    @:native("jsp_reduce_counters")    static function reduceCounters(devices:RawPointer<Int>, ndev:Int, perDevice:RawPointer<cpp.UInt64>, total:RawPointer<cpp.UInt64>, viaRccl:RawPointer<Int>):Int;
    @:native("jsp_pool_create")        static function poolCreate(device:Int, w:Int, h:Int, nbuf:Int):RawPointer<JspPool>;
    @:native("jsp_pool_store_rate")    static function poolStoreRate(p:RawPointer<JspPool>, attempts:RawPointer<Int>):Float;   // diagnostics: what the placement probe of a large pool found
    @:native("jsp_pool_probe_info")    static function poolProbeInfo(p:RawPointer<JspPool>, probeMs:RawPointer<Float>, heldPeak:RawPointer<cpp.UInt64>, holdLimit:RawPointer<cpp.UInt64>):Int;
    @:native("jsp_pool_probe_rates")   static function poolProbeRates(p:RawPointer<JspPool>, rates:RawPointer<Float>, cap:Int):Int;
    @:native("jsp_pool_buffer")        static function poolBuffer(p:RawPointer<JspPool>, i:Int):RawPointer<cpp.Int32>;
    @:native("jsp_pool_destroy")       static function poolDestroy(p:RawPointer<JspPool>):Void;
    @:native("jsp_download")           static function download(deviceFrame:RawConstPointer<cpp.Int32>, host:RawPointer<cpp.Int32>, npixels:SizeT):Int;
    @:native("jsp_display_convert")    static function displayConvert(frame:RawConstPointer<cpp.Int32>, out:RawPointer<cpp.Int32>, w:Int, h:Int, mode:Int, flipRows:Int, stream:RawPointer<cpp.Void>):Int;
    @:native("jsp_frames_differ")      static function framesDiffer(a:RawConstPointer<cpp.Int32>, b:RawConstPointer<cpp.Int32>, firstPixel:SizeT, npixels:SizeT, differ:RawPointer<Int>, stream:RawPointer<cpp.Void>):Int;
}

@:include("jsplayer_amd.h") @:native("jsp_codec") @:structAccess extern class JspCodec {}
@:include("jsplayer_amd.h") @:native("jsp_pool") @:structAccess extern class JspPool {}
#end
