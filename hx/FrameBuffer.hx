// A frame buffer as the codec sees it: width*height Int32 pixels 0x00RRGGBB, bottom-up (Manager.hx:114-118), living in
// HBM.  It stands where the reference has js.lib.Int32Array in IVideoCodec's signatures (IVideoCodec.hx:13,20,24,26):
//
//     typedef FrameData = #if cpp FrameBuffer #else js.lib.Int32Array #end;
//
// Manager's identity tests (`prev_frame == buffers[i]`, Manager.hx:472-475; `res.data_pnt == prev_frame`, :516) keep
// working because FramePool hands out ONE FrameBuffer object per device pointer and NativeCodec maps the pointers the
// C ABI returns back to those objects.
#if cpp
package;

import cpp.RawPointer;

class FrameBuffer {
    public var ptr(default, null):RawPointer<cpp.Int32>;   // device pointer (jsp_pool_buffer)
    public var length(default, null):Int;                   // pixels

    public function new(ptr:RawPointer<cpp.Int32>, length:Int) {
        this.ptr = ptr;
        this.length = length;
    }

    /** The frame's pixels in host memory (display without jsp_display_convert, tests). */
    public function download():haxe.ds.Vector<Int> {
        var out = new haxe.ds.Vector<Int>(length);
        var arr:Array<Int> = cast out.toData();
        JspNative.download(cast ptr, cast cpp.NativeArray.address(arr, 0).raw, length);
        return out;
    }

    /** Key used by FramePool / NativeCodec to find the object that owns a device pointer. */
    public inline function key():haxe.Int64 {
        return untyped __cpp__("(::cpp::Int64)(size_t){0}", ptr);
    }
}

/** The num_buffers + 1 frames Manager allocates (Manager.hx:114-118), in HBM. */
class FramePool {
    public var buffers(default, null):Array<FrameBuffer> = [];
    var pool:RawPointer<JspPool>;
    var byPtr = new Map<String, FrameBuffer>();

    public function new(width:Int, height:Int, count:Int, device:Int = 0) {
        pool = JspNative.poolCreate(device, width, height, count);
        if (pool == null) throw "jsp_pool_create: " + JspNative.lastError().toString();
        for (i in 0...count) {
            var fb = new FrameBuffer(JspNative.poolBuffer(pool, i), width * height);
            buffers.push(fb);
            byPtr.set(haxe.Int64.toStr(fb.key()), fb);
        }
    }

    /** The FrameBuffer that wraps device pointer `p` (null for a null pointer). */
    public function find(p:RawPointer<cpp.Int32>):Null<FrameBuffer> {
        if (p == null) return null;
        var k:haxe.Int64 = untyped __cpp__("(::cpp::Int64)(size_t){0}", p);
        return byPtr.get(haxe.Int64.toStr(k));
    }

    public function dispose():Void {
        if (pool != null) JspNative.poolDestroy(pool);
        pool = null;
        buffers = [];
        byPtr = new Map();
    }
}
#end
